"""CPU: the host-side pieces of bench.py that never see a GPU in the CPU round - the `cpu_baseline` leg of the n1_dual / s2_only workloads
(full-depth as-executed System-2 call of the reference on the CPU oracle, here at a reduced width so it runs in seconds), the argument
defaults the tools rely on, and the schedules of the three cadences (who runs System-2 / System-1 at which step of the period)."""
import numpy as np
import torch

import bench


def test_default_args_match_the_parser(monkeypatch):
    monkeypatch.setattr("sys.argv", ["bench.py"])
    a, d = bench.parse(), bench.default_args(no_cpu_baseline=False, no_variants=False)     # (tools build workloads without the two post-run legs)
    for k, v in vars(a).items():
        assert getattr(d, k) == v, k
    monkeypatch.setattr("sys.argv", ["bench.py", "--workload", "s2_only"])
    assert bench.parse().envs == 7


def test_n1_cpu_baseline_runs_the_as_executed_call_at_reduced_width():
    from internnav_amd import synthetic as S

    qc = dict(S.QWEN_TEST_CFG, v_depth=4, v_fullatt=(1, 3), t_layers=3)
    grid = (1, 8, 8)
    per = 64
    n_img, n_txt = 2, 12
    ids = torch.randint(0, 1000, (1, n_txt + n_img * (per // 4 + 2) + 6))
    o = n_txt
    for _ in range(n_img):
        ids[0, o] = qc["vision_start_id"]
        ids[0, o + 1:o + 1 + per // 4] = qc["image_token_id"]
        ids[0, o + 1 + per // 4] = qc["vision_end_id"]
        o += per // 4 + 2
    for cadence, with_s1 in (("s2_only", False),):
        r = bench.n1_cpu_baseline(qc, [grid] * n_img, n_img * per, ids, n_decode=3, cadence=cadence, with_s1=with_s1, unit="System-2 calls/s")
        assert r["kind"] == "port" and r["value"] > 0 and r["cores"] >= 1
        assert set(r["s2_call_s"]) == {"vit", "prefill", "lm_head_all_positions", "decode_cached_steps", "latents_vit", "latents_prefill"}
        assert r["s2_call_as_executed_s"] >= r["s2_call_without_the_second_forward_s"] > 0
        assert abs(1.0 / r["value"] - sum(r["s2_call_s"].values())) < 0.05 + 0.05 / r["value"]


def _schedule(cadence, B):
    """the schedule arithmetic of bench.N1Dual.__init__ (kept in step with it by construction: same expressions)"""
    P_ = {"nominal": 10, "reference": 8, "s2_only": 1}[cadence]
    mb = [B // P_ + (1 if j < B % P_ else 0) for j in range(P_)]
    st = np.concatenate([[0], np.cumsum(mb)])
    envs_of = lambda j: list(range(int(st[j]), int(st[j]) + mb[j]))  # noqa: E731
    if cadence == "nominal":
        side = [[e for e in range(B) if e not in set(envs_of(j))] for j in range(P_)]
    elif cadence == "reference":
        side = [envs_of((j + P_ // 2) % P_) for j in range(P_)]
    else:
        side = [[] for _ in range(P_)]
    return P_, [envs_of(j) for j in range(P_)], side


def test_cadence_schedules_cover_every_env_as_the_reference_agent_does():
    # nominal: every env runs System-1 every step and System-2 exactly once per 10 steps
    P_, s2, side = _schedule("nominal", 64)
    assert sorted(e for x in s2 for e in x) == list(range(64))
    assert all(sorted(s2[j] + side[j]) == list(range(64)) for j in range(P_))
    # reference agent (internvla_n1_agent.py:210-241, 331-352): per env and 8 steps ONE System-2 call and TWO System-1 calls, 4 steps apart
    P_, s2, side = _schedule("reference", 64)
    for e in range(64):
        s2_steps = [j for j in range(P_) if e in s2[j]]
        s1_steps = sorted([j for j in range(P_) if e in s2[j]] + [j for j in range(P_) if e in side[j]])
        assert len(s2_steps) == 1 and len(s1_steps) == 2 and (s1_steps[1] - s1_steps[0]) == 4
    # System-2 only: everybody, every step
    P_, s2, side = _schedule("s2_only", 7)
    assert s2 == [list(range(7))] and side == [[]]


def test_bench_rank_logic_world8_gloo_host_stub():
    """`python bench.py --gpus 8 --workload host_stub` on CPU: the file's multi-rank path end to end without a GPU -
    self-spawn of 8 ranks through torch.distributed.run on 127.0.0.1, per-rank core pinning, the per-step all-gather of the [64, 4] int32
    action table with its content asserts (own slice, valid ids, same bytes on every rank), max-over-ranks timing, rank 0 alone in the
    post-run section - around the real per-step host work (64 x traj_to_actions per rank). One JSON line, whole-job value."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "8", "--workload", "host_stub", "--steps", "4", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=120, cwd=str(root))       # (120 s: 8 ranks must come up and finish well inside the driver's limits)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["gloo_ranks"] == 8 and d["dist"]["backend"] == "gloo" and d["dist"]["ranks"] == 8 and len(d["dist"]["setup_s_per_rank"]) == 8 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "dp8" and d["config"]["envs_per_gpu"] == 64
    # whole-job aggregate: 8 ranks x 64 envs x steps / max-over-ranks time
    assert abs(d["value"] - 8 * 64 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]
    print(f"8 ranks x 64 traj_to_actions per step on {len(os.sched_getaffinity(0))} host cores: {d['ms_per_step']:.1f} ms per step "
          f"({d['config']['host_cores_per_rank']} core(s) per rank)")
