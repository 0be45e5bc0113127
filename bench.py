#!/usr/bin/env python
"""Benchmark of the MI355X policy engine: policy steps / sec / node (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--envs 64] [--workload n1_dual|navdp_s1]
  N > 1 is launched by the driver as  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Workloads (config.workload):
  n1_dual_b64 (default) - the configuration BASELINE.json's metric is quoted on: InternVLA-N1 full dual system (ViT + Qwen2.5-VL-7B
      System-2 @ 1 Hz + NextDiT System-1 @ 10 Hz), 64 parallel episodes per GPU (512 per 8-GPU node), 4 x RGB frames (392x392 after
      the HF processor = 784 patches each) + 64-token instruction -> S = 920 prompt tokens, 8 greedy answer tokens + 4 latent queries
      per System-2 call, 2 look-down frames of 224x224 + 32 samples x 10 flow-matching steps per System-1 call. One bench step = ONE
      policy step of every env of the rank: System-1 for all 64 envs, System-2 for the 6-7 envs whose plan expires this step
      (every env exactly once per 10 steps), i.e. 64 policy steps at the nominal 1 : 10 cadence.
  navdp_s1_b64 - BASELINE config #2: NavDP System-1 only (NavDPNet, 10 DDPM steps x 32 samples + critic ranking, 8 RGB + 1 depth frame).
Weights are seeded random at the true architecture shapes (7.6 B + 0.68 B + S1 parameters, drawn on the device; no checkpoint is
available offline), inputs synthetic and resident in HBM before the timed region. Episodes are independent, so ranks shard envs
with no data-path exchange ("scaling": "weak"); the only collective is the all_gather of per-env actions over RCCL/xGMI per step.

Timed region: W warm-up steps, then exactly K steps bracketed by barrier + device synchronise, max over ranks. Each engine call
replays a hipGraph: System-2 ViT + prefill of the 6- or 7-env micro-batch; then its decode + latent-query passes on the main stream
concurrently with System-1 of the 57-58 other envs on a side stream; then System-1 of the micro-batch's envs (same results as the
single-stream order, checked before the timed region). Noise draw, micro-batch gather/scatter, D2H of the trajectories and the host
post-processing (traj_to_actions; for the side-stream envs it runs while the GPU finishes the main stream) are inside the step.
After the timed region rank 0 runs ONE instrumented eager pass with per-launch HIP events (ina_prof_*) to attribute time and
algorithmic FLOPs to kernel classes for the "roofline" object, and (N = 1 only) times the CPU oracle on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL / cross-process tensors) on this driver

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16 peak of MI355X (guides/MI355X_MICROARCH.md: ~2.5 PF dense)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs", type=int, default=None, help="environments per GPU (default 64; 7 for --workload s2_only)")
    ap.add_argument("--workload", default="n1_dual", choices=["n1_dual", "s2_only", "navdp_s1", "unet1d_s1", "sft", "host_stub"],
                    help="n1_dual = the BASELINE metric's configuration (default); s2_only = config #3 (System-2 calls/s); navdp_s1 = config #2; "
                         "unet1d_s1 = the diffusion-policy UNet head; sft = config #5 (the SFT step, bench_sft.py: its own flags pass through); "
                         "host_stub = no GPU: the multi-rank logic (self-spawn, pinning, action all-gather over gloo) around the real per-step host work")
    ap.add_argument("--cadence", choices=["nominal", "reference"], default="nominal",
                    help="n1_dual: nominal = 1 S2 : 10 S1 per env (the BASELINE metric, default); reference = the agent's own schedule, (1 S2 + 2 S1) per 8 "
                         "actions and env (SURVEY.md 8d, internvla_n1_agent.py:210-241)")
    ap.add_argument("--num-history", type=int, default=3,
                    help="n1_dual / s2_only: history frames per System-2 prompt besides the current frame (3 = the metric's 4 frames; the reference's "
                         "evaluation harness uses 8, scripts/eval/configs/habitat_dual_system_cfg.py:10-12)")
    ap.add_argument("--lookdown", action="store_true",
                    help="n1_dual / s2_only: the look-down turn's prompt - previous turn + answer + the UN-RESIZED 640x480 look-down frame "
                         "(internvla_n1_policy.py:113-116,140)")
    ap.add_argument("--dit-ffn", type=int, default=1536, choices=[1536, 1024],
                    help="n1_dual: FFN width of the NextDiT trajectory head - 1536 = the reference's `LuminaFeedForward(dim, inner_dim=4 * dim)` under its "
                         "pinned diffusers 0.33.1 (default), 1024 = the same call under diffusers <= 0.32 (what rounds 1-5 timed)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true",
                    help="n1_dual, N = 1: skip the 10-step timings of the exact schedule variants (--prefix-kv, --s2-every 2) and of the other FFN width "
                         "that follow the headline measurement and are reported under `variants` of the same JSON line")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay (debug)")
    ap.add_argument("--no-overlap", action="store_true", help="n1_dual: run System-2 and System-1 back to back on one stream")
    ap.add_argument("--vit-cache", action="store_true",
                    help="n1_dual: per-frame ViT cache variant - the first history frame of every env (frame 0, present in every np.linspace history "
                         "sample of the reference) comes from the cache; algorithmic FLOPs are accounted accordingly")
    ap.add_argument("--prefix-kv", action="store_true",
                    help="n1_dual: prefix-KV reuse variant - the K/V of system prompt + instruction + first history frame (296 of the 920 prompt "
                         "tokens, identical between the System-2 calls of an episode) come from a per-env cache; the call encodes 3 of 4 frames and "
                         "prefills 624 tokens per env. Exact (causal mask); algorithmic FLOPs are accounted accordingly. Reported next to the headline.")
    ap.add_argument("--s2-every", type=int, default=1,
                    help="n1_dual, nominal cadence: run the System-2 micro-batch only on every E-th step (E = 2: 12-13 envs every other step instead of 6-7 every "
                         "step). Every env still runs System-2 once per 10 steps and System-1 every step; the decode / latent-query passes stream the "
                         "15 GB of decoder weights 9 times per TWO steps instead of per step. Trades step-latency uniformity for throughput; reported next to the default.")
    ap.add_argument("--no-split-prefill", action="store_true",
                    help="n1_dual: System-2 prefill as ONE launch sequence instead of two half micro-batches on two streams")
    ap.add_argument("--no-frag-weights", action="store_true", help="n1_dual: prefill GEMMs without the fragment-ordered weight copies (tile config 39 / 18 instead of 40)")
    ap.add_argument("--no-fuse-decode-rope", action="store_true", help="n1_dual: the single-token passes with the rope + KV-append launch of their own (6 launches per layer instead of 5)")
    ap.add_argument("--no-row-chain", action="store_true",
                    help="n1_dual: the round-4 launches for the row-local part of the NextDiT blocks (GEMM, norm, GEMM) instead of the row-chain kernel")
    # (round-3 schedule experiments - stream priorities, System-1 started with the prefill, early look-down encoding, a split side-stream
    #  call - were all measured neutral or negative, profiles/r03d/e/f/u_*; their switches are gone)
    a, rest = ap.parse_known_args()
    if rest and a.workload != "sft":
        ap.error(f"unrecognized arguments: {' '.join(rest)}")
    a.rest = rest
    if a.envs is None:
        a.envs = 7 if a.workload == "s2_only" else 64
    return a


def default_args(**kw):
    """the parsed defaults as a namespace (tools/ construct workloads without a command line)"""
    a = argparse.Namespace(gpus=1, steps=20, warmup=3, envs=64, workload="n1_dual", cadence="nominal", num_history=3, lookdown=False, no_cpu_baseline=True, dit_ffn=1536,
                           no_graph=False, no_overlap=False, vit_cache=False, prefix_kv=False, no_split_prefill=False, s2_every=1, no_row_chain=False,
                           no_frag_weights=False, no_fuse_decode_rope=False, no_variants=True, rest=[])
    for k, v in kw.items():
        assert hasattr(a, k), k
        setattr(a, k, v)
    return a


def calibration_gemm(dev, seconds: float = 0.5):
    """sustained TF/s of THIS box on one fixed compute-bound launch (8192^3 bf16 through the library's own tiled GEMM), so the line tells
    box-to-box clock / power spread (265-281 steps/s across round-2 boxes on unchanged code) apart from code changes."""
    from internnav_amd import ops

    x = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    w = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16) * 0.01
    out = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.linear(x, w, out=out)
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            ops.linear(x, w, out=out)
        torch.cuda.synchronize()
        n += 10
    dt = time.perf_counter() - t0
    return {"gemm_8192_tflops": round(n * 2 * 8192 ** 3 / dt / 1e12, 1), "launches": n, "seconds": round(dt, 2)}


# ------------------------------------------------------------------------------------------------------------ workloads
class NavDPS1:
    """BASELINE config #2."""

    def __init__(self, a, dev, rank):
        from internnav_amd import flops, synthetic
        from internnav_amd.navdp import NavDPNet

        self.cfg = cfg = synthetic.NAVDPNET_CFG
        self.B = B = a.envs
        self.name = f"navdp_s1_b{B}"
        self.desc = {"policy": "NavDPNet (BASELINE config #2: NavDP System-1 only)", "samples_per_env": cfg["sample_num"],
                     "ddpm_steps": cfg["num_train_timesteps"], "frames_per_env": f"{cfg['memory_size']} rgb + 1 depth @224x224"}
        self.net = NavDPNet(synthetic.navdpnet_state_dict(seed=0), cfg, dev, max_envs=B)
        self.g = g = torch.Generator(device=dev).manual_seed(1000 * rank + 7)
        self.inp = dict(
            goal=torch.randn(B, 3, device=dev, generator=g) * 3.0,
            images=torch.rand(B, cfg["memory_size"], 224, 224, 3, device=dev, generator=g),
            depths=torch.rand(B, 1, 224, 224, 1, device=dev, generator=g) * 5.0,
            x_init=torch.randn(B, cfg["sample_num"], cfg["predict_size"], 3, device=dev, generator=g),
            step_noise=torch.randn(cfg["num_train_timesteps"], B, cfg["sample_num"], cfg["predict_size"], 3, device=dev, generator=g))
        self.f_alg = flops.navdpnet_flops_per_env(cfg)["total"]
        self.graph = None
        self.action_shape = (B, 8, cfg["predict_size"], 3)

    def _call(self, goal, images, depths, x_init, step_noise):
        return self.net.predict_pointgoal_batch_action_vel(goal, images, depths, x_init, step_noise)

    def capture(self):
        from internnav_amd import runtime

        self.graph = runtime.GraphedCall(self._call, self.inp)

    def step(self, i):
        self.inp["x_init"].normal_(generator=self.g)
        self.inp["step_noise"].normal_(generator=self.g)
        neg, pos = self.graph() if self.graph else self._call(**self.inp)
        self.last_out = pos
        return pos

    def instrumented(self):
        self._call(**self.inp)

    def step_output_for_check(self):
        return self.last_out

    def cpu_baseline(self):
        from internnav_amd import synthetic
        from oracle import navdp as o_navdp  # cpu_baseline leg only

        cores = min(64, os.cpu_count() or 1)
        torch.set_num_threads(cores)
        sd = synthetic.navdpnet_state_dict(0)
        inp = synthetic.navdpnet_inputs(1, 0)
        fn = lambda: o_navdp.navdpnet_pointgoal(sd, inp["goal"], inp["images"], inp["depths"], inp["x_init"], inp["step_noise"], self.cfg)  # noqa: E731
        t32, t16 = _median_time(fn, 3), _median_time(fn, 3, autocast=True)
        dt = min(t32, t16)
        return {"value": round(1.0 / dt, 4), "unit": "policy steps/s", "cores": cores, "cpu": _cpu_model(), "kind": "port",
                "seconds": {"fp32": round(t32, 2), "bf16_autocast": round(t16, 2)},
                "sample": "1 env x 1 policy step (9 ViT-S frames + 10 DDPM steps x 32 samples + critic), torch CPU, batch-1 as the reference executes; "
                          "1 warm-up + median of 3 runs each for fp32 and bf16-autocast, value = the faster"}


class UNet1DS1:
    """diffusion-policy ConditionalUnet1D + 10 DDIM steps as a System-1 head, 64 envs x 32 samples per call."""

    def __init__(self, a, dev, rank):
        from internnav_amd import flops, synthetic
        from internnav_amd.unet1d import UNet1DHead

        self.cfg = cfg = synthetic.UNET1D_CFG
        self.B = B = a.envs
        self.name = f"unet1d_s1_b{B}"
        self.desc = {"policy": "ConditionalUnet1D (diffusion-policy, down [256,512,1024], kernel 5, FiLM) + DDIM", "samples_per_env": cfg["sample_num"],
                     "ddim_steps": cfg["num_inference_steps"], "condition": f"one {cfg['global_cond_dim']}-d vector per env"}
        self.net = UNet1DHead(synthetic.materialize(synthetic.unet1d_spec(cfg), 0), cfg, dev, max_envs=B)
        self.g = g = torch.Generator(device=dev).manual_seed(1000 * rank + 7)
        self.inp = dict(global_cond=torch.randn(B, cfg["global_cond_dim"], device=dev, generator=g),
                        x_init=torch.randn(B, cfg["sample_num"], cfg["predict_size"], cfg["input_dim"], device=dev, generator=g))
        self.f_alg = flops.unet1d_flops_per_env(cfg)["total"]
        self.graph = None
        self.action_shape = (B, cfg["sample_num"], cfg["predict_size"], cfg["input_dim"])

    def _call(self, global_cond, x_init):
        return self.net.sample_traj(global_cond, x_init)

    def capture(self):
        from internnav_amd import runtime

        self.graph = runtime.GraphedCall(self._call, self.inp)

    def step(self, i):
        self.inp["x_init"].normal_(generator=self.g)
        out = self.graph() if self.graph else self._call(**self.inp)
        self.last_out = out
        return out

    def instrumented(self):
        self._call(**self.inp)

    def step_output_for_check(self):
        return self.last_out

    def cpu_baseline(self):
        from internnav_amd import synthetic
        from oracle import unet1d as o_u  # cpu_baseline leg only

        cores = min(64, os.cpu_count() or 1)
        torch.set_num_threads(cores)
        sd = synthetic.materialize(synthetic.unet1d_spec(self.cfg), 0)
        inp = synthetic.unet1d_inputs(1, 0, self.cfg)
        fn = lambda: o_u.ddim_sample(sd, inp["global_cond"], inp["x_init"], self.cfg["num_train_timesteps"], self.cfg["num_inference_steps"])  # noqa: E731
        t32, t16 = _median_time(fn, 3), _median_time(fn, 3, autocast=True)
        dt = min(t32, t16)
        return {"value": round(1.0 / dt, 4), "unit": "policy steps/s", "cores": cores, "cpu": _cpu_model(), "kind": "port",
                "seconds": {"fp32": round(t32, 2), "bf16_autocast": round(t16, 2)},
                "sample": "1 env x 1 policy step (32 samples x 10 DDIM steps of the UNet), torch CPU; 1 warm-up + median of 3 runs each for fp32 and bf16-autocast"}


class HostStub:
    """`--workload host_stub` (no GPU, backend gloo): the rank logic of this file - self-spawned ranks, core pinning, the per-step all-gather of
    the [envs, 4] int32 action table with its content checks, max-over-ranks timing, rank 0 alone in the post-run section while the others wait
    at the barrier - around the REAL per-step host work of the n1_dual workload: `traj_to_actions` of 64 trajectories [32, 32, 3] per rank
    (vln_utils.py:36-136, what the reference does per env and step). It answers the question a 1-GPU box cannot: do eight ranks' Python
    post-processing loops limit the step on one host? The engine time of a step is NOT simulated: `ms_per_step` here is
    the host floor under the GPU time (2.9 ms at one rank on the GPU box; the side-stream envs' share runs under the GPU tail anyway)."""

    def __init__(self, a, dev, rank):
        from internnav_amd.policy import traj_to_actions

        self.traj_to_actions, self.B = traj_to_actions, a.envs
        self.name, self.unit = f"host_stub_b{a.envs}", "policy steps/s"
        g = torch.Generator().manual_seed(1000 * rank + 7)
        # random-walk-like trajectories of the System-1 output shape (x4-scaled waypoint increments, 32 samples x 32 steps)
        self.traj = torch.randn(8, self.B, 32, 32, 3, generator=g) * 0.5 + torch.tensor([0.6, 0.0, 0.0])
        self.actions = torch.zeros(self.B, 4, dtype=torch.int32)
        self.action_shape = (self.B, 4)
        self.f_alg = 0.0
        self.desc = {"policy": "none (host stub): traj_to_actions of 64 x [32, 32, 3] trajectories per rank and step + the action all-gather over gloo"}

    def capture(self):
        pass

    def step(self, i):
        t = self.traj[i % self.traj.shape[0]]
        acts = np.zeros((self.B, 4), dtype=np.int32)
        for b in range(self.B):
            al = [x for x in self.traj_to_actions(t[b]) if x != 0][:4]
            acts[b, :len(al)] = al
        self.actions.copy_(torch.from_numpy(acts))
        return self.actions

    def step_output_for_check(self):
        return self.actions


class N1Dual:
    """InternVLA-N1 dual system (the configuration the BASELINE metric is quoted on) - cadence and System-2 prompt shape are parameters:

      cadence 'nominal'   (default, BASELINE): S2 @ 1 Hz : S1 @ 10 Hz - every env runs System-1 every step and System-2 once per 10 steps
      cadence 'reference' (internvla_n1_agent.py:210-241 + :331-352): per env and 8 actions ONE System-2 call and TWO System-1
                          calls (each System-1 plan yields 4 discrete actions, sys2_max_forward_step = 8); the other 6 steps pop queued
                          actions on the host (step_no_infer). Per bench step: System-2 for 8 envs, System-1 for those 8 + the 8 envs half a
                          period away, 64 actions out.
      cadence 's2_only'   (BASELINE config #3): every env of the (small) batch runs System-2 every step, no System-1; unit = System-2 calls/s

      --num-history H   H history frames + the current frame per System-2 prompt (default 3 -> the metric's 4 frames; the reference's
                        evaluation harness runs 8, scripts/eval/configs/habitat_dual_system_cfg.py:10-12)
      --lookdown        the look-down turn: the prompt of the previous turn + its answer + the UN-RESIZED 640x480 look-down frame
                        (internvla_n1_policy.py:113-116,140: 46 x 34 patches = 391 tokens) - the longest prompt the harness produces
    """

    GRID, N_INSTR, N_DECODE = (1, 28, 28), 64, 8

    def __init__(self, a, dev, rank):
        from internnav_amd import flops, synthetic
        from internnav_amd.policy import InternVLAN1ForCausalLM, traj_to_actions
        from internnav_amd.preprocess import smart_resize

        self.traj_to_actions = traj_to_actions
        self.a = a
        self.cadence = "s2_only" if a.workload == "s2_only" else a.cadence
        self.dev, self.B = dev, a.envs
        B = self.B
        self.N_IMG = a.num_history + 1
        self.lookdown = bool(a.lookdown)
        tag = ("" if self.cadence == "nominal" else f"_{self.cadence}") + ("" if a.num_history == 3 else f"_h{a.num_history}") + ("_lookdown" if self.lookdown else "")
        self.name = (f"n1_s2_only_b{B}" if self.cadence == "s2_only" else f"n1_dual_b{B}") + tag.replace("_s2_only", "")
        qcfg, scfg = synthetic.QWEN_N1_CFG, synthetic.N1_NEXTDIT_VARIANTS[f"ffn{int(getattr(a, 'dit_ffn', 1536))}"]
        self.qcfg, self.scfg = qcfg, scfg
        per = self.GRID[1] * self.GRID[2]
        hb, wb = smart_resize(480, 640)                                   # the look-down frame enters the HF processor at camera size
        self.LD_GRID = (1, hb // 14, wb // 14)
        per_ld = self.LD_GRID[1] * self.LD_GRID[2]
        self.grids_seq = [self.GRID] * self.N_IMG + ([self.LD_GRID] if self.lookdown else [])
        self.pv_rows_seq = self.N_IMG * per + (per_ld if self.lookdown else 0)
        # prompt layout (HF chat template shape): 34 template tokens | 64 instruction tokens | N x (<vs> 196 x <img> <ve>) | 30 tail tokens
        # look-down turn: + 16 tokens (the previous answer and the next turn's template) | <vs> 391 x <img> <ve> | 8 tail tokens
        self.S = 34 + self.N_INSTR + self.N_IMG * (per // 4 + 2) + 30 + ((16 + per_ld // 4 + 2 + 8) if self.lookdown else 0)
        # ---- schedule: per step j of the period, the envs of the System-2 micro-batch (contiguous) and the envs whose System-1 call does
        # not depend on it (side stream)
        if self.cadence == "nominal":
            self.PERIOD = 10
        elif self.cadence == "reference":
            self.PERIOD = 8
        else:
            self.PERIOD = 1
        P_ = self.PERIOD
        self.s2_every = max(1, int(getattr(a, "s2_every", 1)))
        assert self.s2_every == 1 or (self.cadence == "nominal" and P_ % self.s2_every == 0), "--s2-every needs the nominal cadence and must divide its period of 10"
        slots = [j for j in range(P_) if j % self.s2_every == 0]          # the steps of a period that carry a System-2 micro-batch
        per_slot = [B // len(slots) + (1 if k < B % len(slots) else 0) for k in range(len(slots))]
        self.mb = [per_slot[slots.index(j)] if j in slots else 0 for j in range(P_)]   # micro-batch sizes, sum = B (0 = a System-1-only step)
        self.mb_start = np.concatenate([[0], np.cumsum(self.mb)])
        assert min(per_slot) >= 1, f"--envs {B} is smaller than the number of System-2 steps per period ({len(slots)})"
        mmax = max(self.mb)
        if self.s2_every > 1:
            tag += f"_s2every{self.s2_every}"
            self.name += f"_s2every{self.s2_every}"

        def envs_of(j):
            return list(range(int(self.mb_start[j]), int(self.mb_start[j]) + self.mb[j]))
        if self.cadence == "nominal":
            side = [[e for e in range(B) if e not in set(envs_of(j))] for j in range(P_)]
        elif self.cadence == "reference":
            side = [envs_of((j + P_ // 2) % P_) for j in range(P_)]
        else:
            side = [[] for _ in range(P_)]
        self.idxA_host = side
        self.with_s1 = self.cadence != "s2_only"
        s1_max = max(len(side[j]) + self.mb[j] for j in range(P_)) if self.with_s1 else 1     # (the single-stream schedule runs both groups in one call)
        spec = synthetic.n1_full_spec(qcfg, "nextdit_async", s1_cfg=scfg)
        weights = synthetic.LazyDeviceWeights(spec, dev, seed=0)
        S_max = (self.S + self.N_DECODE + 8 + 63) // 64 * 64
        self.model = InternVLAN1ForCausalLM(weights, qcfg, "nextdit_async", scfg, device=dev, max_envs=(B if self.cadence == "nominal" else s1_max),
                                            max_seq_len=max(1024, S_max), max_patches=mmax * self.pv_rows_seq, max_s2_seqs=mmax)
        if getattr(a, "no_row_chain", False):
            self.model.s1.row_chain = False
        if getattr(a, "no_split_prefill", False):
            self.model.qwen.split_prefill = False
        if getattr(a, "no_frag_weights", False):
            self.model.qwen.drop_frag_weights()
        if getattr(a, "no_fuse_decode_rope", False):
            self.model.qwen.fuse_decode_rope = False
        g = self.g = torch.Generator(device=dev).manual_seed(1000 * rank + 7)
        lim = qcfg["image_token_id"] - 16
        ids = torch.randint(0, lim, (B, self.S), device=dev, generator=g)
        o = 34 + self.N_INSTR
        for k in range(self.N_IMG):
            ids[:, o] = qcfg["vision_start_id"]
            ids[:, o + 1:o + 1 + per // 4] = qcfg["image_token_id"]
            ids[:, o + 1 + per // 4] = qcfg["vision_end_id"]
            o += per // 4 + 2
        if self.lookdown:
            o += 30 + 16
            ids[:, o] = qcfg["vision_start_id"]
            ids[:, o + 1:o + 1 + per_ld // 4] = qcfg["image_token_id"]
            ids[:, o + 1 + per_ld // 4] = qcfg["vision_end_id"]
        self.ids = ids
        self.grid = torch.tensor([list(gr) for gr in self.grids_seq])
        self.images_dp = torch.rand(B, 2, 224, 224, 3, device=dev, generator=g).to(torch.bfloat16)
        # raw camera frames (the metric's input: N x 640x480 RGB per env) resident in HBM; every step runs them through the
        # bit-exact device pre-processor (PIL bicubic 640x480 -> 384x384 -> 392x392, rescale / normalise / patchify for the System-2
        # micro-batch; 640x480 -> 224x224, / 255 for the System-1 look-down pair = frames 0 and N-1), one batched launch per stage
        self.raw = True          # (the round-1 boundary - resident pixel_values / 224x224 frames - is gone: the timed step starts at raw camera frames)
        if self.raw:
            from internnav_amd.preprocess import FramePreprocessor

            self.pre = FramePreprocessor(dev, resize_w=384, resize_h=384)
            self.frames_u8 = torch.randint(0, 256, (B, self.N_IMG + (1 if self.lookdown else 0), 480, 640, 3), device=dev, generator=g, dtype=torch.uint8)
            self.s1_sel = torch.tensor([0, self.N_IMG - 1], device=dev)
        else:
            self.pixel_values = torch.randn(B, self.N_IMG * per, 1176, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16)
        self.latent_table = torch.randn(B, qcfg["n_query"], qcfg["t_hidden"], device=dev, generator=g).to(torch.bfloat16)
        self.x_init = torch.randn(B, scfg["sample_num"], scfg["predict_size"], 3, device=dev, generator=g)
        self.desc = {"policy": "InternVLA-N1 dual system (Qwen2.5-VL-7B S2 + NextDiT-async S1)",
                     "cadence": {"nominal": "nominal: 1 S2 : 10 S1 per env (S2 @ 1 Hz, S1 @ 10 Hz)",
                                 "reference": "reference agent: (1 S2 + 2 S1) per 8 actions and env (internvla_n1_agent.py:210-241,331-352); 6 of 8 steps pop queued actions",
                                 "s2_only": "System-2 only (BASELINE config #3): every env runs one System-2 call per step; value = System-2 calls/s"}[self.cadence],
                     "input": ("raw uint8 640x480 RGB frames resident in HBM, device pre-processing (PIL-exact resize, HF rescale/normalise/patchify) inside the timed step"
                               if self.raw else "pre-processed pixel_values / 224x224 frames resident in HBM"),
                     "s2": f"{self.N_IMG} frames x 784 patches" + (f" + the un-resized look-down frame ({per_ld} patches)" if self.lookdown else "") +
                           f" + {self.N_INSTR}-token instruction, S={self.S}, {self.N_DECODE} greedy tokens + 4 latent queries",
                     "s1_dit": (f"NextDiT {scfg['dit_layers']} blocks x dim {scfg['dit_dim']}, {scfg['dit_heads']} heads, FFN {scfg['dit_ffn']} "
                                + ("(= the reference's LuminaFeedForward(dim, inner_dim=4*dim) under its pinned diffusers 0.33.1; unverified against a released checkpoint)"
                                   if scfg["dit_ffn"] == 1536 else "(= the diffusers <= 0.32 convention of LuminaFeedForward; rounds 1-5 timed this width)")) if self.with_s1 else "none",
                     "s1": ("2 look-down frames @224x224, 32 samples x 10 flow-matching steps" + ("" if getattr(a, "no_row_chain", False) else
                            ("; row-local chain of every DiT block in two launches (dit_rowchain, 128-row panels)" if scfg["dit_ffn"] <= 1024 else
                             "; attn2.to_out + norms + SwiGLU of every DiT block in one launch (dit_rowchain, 128-row panels), linear_2 and the next projection as GEMM + norm launches"))) if self.with_s1 else "none",
                     "s2_microbatches_per_period": self.mb, "s2_every": self.s2_every, "s1_side_stream_envs_per_step": sorted(set(len(x) for x in side))}
        self.desc["s2_prefill"] = ("two half micro-batches on two streams (fork / join inside the captured launch sequence), GEMM tiles selected in the shared-tail mode (force_cfg = -1)"
                                   if self.model.qwen.split_prefill else "one launch sequence")
        if not getattr(a, "no_s1_merge_images", False) and self.cadence == "nominal" and self.with_s1 and not getattr(a, "no_overlap", False) and not a.no_graph:
            self.desc["s1_images"] = "look-down pairs of all 64 envs encoded in ONE pass inside the side-stream System-1 call; the call of the System-2 envs starts at the projected latents"
        self.vit_cache = bool(getattr(a, "vit_cache", False)) and self.raw
        self.prefix_kv = bool(getattr(a, "prefix_kv", False)) and self.raw
        assert not (self.vit_cache and self.prefix_kv), "--prefix-kv already covers frame 0 (its tokens are cached K/V): use one of the two"
        assert not ((self.vit_cache or self.prefix_kv) and self.lookdown), "the exact-reuse variants are benchmarked on the plain prompt shape"
        self.prefix_len = (34 + self.N_INSTR + per // 4 + 2) if self.prefix_kv else 0     # template + instruction + <vs> frame 0 <ve>
        n_fresh = self.N_IMG - 1 if (self.vit_cache or self.prefix_kv) else self.N_IMG
        fresh_grids = [self.GRID] * n_fresh + ([self.LD_GRID] if self.lookdown else [])
        self.pv_rows_fresh = n_fresh * per + (per_ld if self.lookdown else 0)
        f2 = flops.s2_call_flops(self.S, fresh_grids, self.N_DECODE, qcfg, prefix_len=self.prefix_len)   # cached frames / tokens cost no FLOPs
        f1 = flops.nextdit_s1_flops_per_env(scfg)
        self.f_alg = {"nominal": f1["total"] + f2["total"] / 10, "reference": (f2["total"] + 2 * f1["total"]) / 8, "s2_only": f2["total"]}[self.cadence]
        self.f_parts = {"s1_per_env": f1["total"], "s2_per_call": f2["total"]}
        self.unit = "System-2 calls/s" if self.cadence == "s2_only" else "policy steps/s"
        # static S2 buffers per micro-batch size
        q = self.model.qwen
        self.s2 = {}
        if self.vit_cache:
            # frame-0 embeddings of every env, computed once (they would have been produced by the env's first System-2 call)
            self.emb0 = torch.empty(B, per // 4, qcfg["t_hidden"], dtype=torch.bfloat16, device=dev)
            for lo in range(0, B, mmax):
                hi = min(B, lo + mmax)
                pv0, _ = self.pre.qwen_pixel_values(self.frames_u8[lo:hi, 0].contiguous())
                emb, inv = q.vision(pv0, [self.GRID] * (hi - lo))
                self.emb0[lo:hi].copy_(emb[torch.from_numpy(inv).to(dev).long()].view(hi - lo, per // 4, -1))
            self.desc["vit_cache"] = f"frame 0 of every env from the per-frame ViT cache ({n_fresh} of {self.N_IMG} frames encoded per System-2 call)"
        if self.prefix_kv:
            # prefix K/V of every env, computed once by a prefill of the prefix alone (they would have been left behind by the env's
            # previous System-2 call): bf16 [B, layers, prefix_len, 1024] = 17 MB per env
            pl = self.prefix_len
            self.kv_prefix = torch.empty(B, qcfg["t_layers"], pl, q.kv_w, dtype=torch.bfloat16, device=dev)
            for lo in range(0, B, mmax):
                hi = min(B, lo + mmax)
                pv0, _ = self.pre.qwen_pixel_values(self.frames_u8[lo:hi, 0].contiguous())
                q.prefill(ids[lo:hi, :pl].cpu(), pv0, torch.cat([self.grid[:1]] * (hi - lo)))
                for k in range(hi - lo):
                    self.kv_prefix[lo + k].copy_(q.export_prefix_kv(k, pl))
            self.desc["prefix_kv"] = (f"K/V of the first {pl} prompt tokens (template + instruction + frame 0) of every env from the prefix cache: "
                                      f"{self.S - pl} tokens prefilled and {n_fresh} of {self.N_IMG} frames encoded per System-2 call")
        for m in sorted(set(self.mb) - {0}):
            cache0 = torch.empty(m, per // 4, qcfg["t_hidden"], dtype=torch.bfloat16, device=dev) if self.vit_cache else None
            cached = [c for k in range(m) for c in ([cache0[k]] + [None] * (self.N_IMG - 1))] if self.vit_cache else None
            P = q.plan(ids[:m].cpu(), torch.cat([self.grid] * m), n_decode=self.N_DECODE, with_latents=True, cached_embeds=cached,
                       prefix_len=self.prefix_len)
            self.s2[m] = dict(P=P, cache0=cache0, pv=torch.empty(m * self.pv_rows_fresh, 1176, dtype=torch.bfloat16, device=dev),
                              toks=torch.zeros(m, self.N_DECODE, dtype=torch.int32, device=dev),
                              lat=torch.zeros(m, qcfg["n_query"], qcfg["t_hidden"], dtype=torch.bfloat16, device=dev), graph=None)
        self.s1_graph = None
        self.action_shape = (B, 4)
        self.actions = torch.zeros(B, 4, dtype=torch.int32, device=dev)
        self.queue = [[] for _ in range(B)]           # reference cadence: the actions of an env's current System-1 plan not yet executed
        # Two-stream schedule (same results, different order): System-1 of the envs that do NOT run System-2 this step is independent
        # of it and is launched on a side stream, concurrently with the decode + latent-query passes of the System-2 micro-batch (chains of
        # short weight-streaming kernels that leave the MFMA pipes idle); System-1 of the System-2 envs follows System-2 on the main
        # stream, on a second small engine instance (own workspace).
        self.overlap = self.with_s1 and not getattr(a, "no_overlap", False) and not a.no_graph
        if self.overlap:
            from internnav_amd.nextdit import NextDiTSystem1
            from internnav_amd.policy import _Prefixed

            self.s1_small = NextDiTSystem1(_Prefixed(weights, "model."), scfg, dev, max_envs=mmax)
            # the look-down pairs of the System-2 envs go through the side stream's encoder pass (DINOv2, MemoryEncoder, QFormer over all 64 envs at
            # once) and only their 32 memory tokens travel to the small engine: the latency-bound encoder chain (~180 launches) leaves the main
            # chain, 302.5 / 303.0 -> 305.6 policy steps/s on one box (profiles/r04o_bench_s1_variants.log; the row-norm epilogue in the small
            # engine alone: 300.5, not adopted)
            self.merge_images = not getattr(a, "no_s1_merge_images", False) and self.cadence == "nominal"
            self.side = torch.cuda.Stream(device=dev)
            nA = max(len(x) for x in side)
            self.latA, self.imgA, self.xA = (torch.empty((nA + (mmax if t is self.images_dp else 0),) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
                                             for t in (self.latent_table, self.images_dp, self.x_init))     # imgA: side envs | System-2 envs (merged encoder pass)
            self.latB, self.imgB, self.xB = (torch.empty((mmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
                                             for t in (self.latent_table, self.images_dp, self.x_init))
            self.idxA = [torch.tensor(x, device=dev, dtype=torch.long) for x in side]
            self.traj = torch.zeros(B, scfg["sample_num"], scfg["predict_size"], 3, device=dev)
            self.hostA = torch.empty(nA, scfg["sample_num"], scfg["predict_size"], 3).pin_memory()
            self.hostB = torch.empty(mmax, scfg["sample_num"], scfg["predict_size"], 3).pin_memory()
            self.gA, self.gB, self.gP, self.gD, self.gAimg = {}, {}, {}, {}, {}
            self.ev, self.ev_img = torch.cuda.Event(), torch.cuda.Event()

    # ---- ingest: raw frames -> engine inputs (inside the timed step)
    def _ingest_s2(self, lo, m, dst):
        """System-2 images of envs [lo, lo + m): raw frames -> pixel_values of the micro-batch (or the round-1 resident tensor)."""
        N = self.N_IMG
        if self.prefix_kv or self.vit_cache:
            if self.prefix_kv:
                self.model.qwen.import_prefix_kv_batch(self.kv_prefix[lo:lo + m])       # 28 strided device copies, inside the timed step
            else:
                self.s2[m]["cache0"].copy_(self.emb0[lo:lo + m])
            pv, _ = self.pre.qwen_pixel_values(self.frames_u8[lo:lo + m, 1:].reshape(m * (N - 1), 480, 640, 3))
            dst.copy_(pv)
        elif self.raw and not self.lookdown:
            pv, _ = self.pre.qwen_pixel_values(self.frames_u8[lo:lo + m].reshape(m * N, 480, 640, 3))
            dst.copy_(pv)
        elif self.raw:
            # history / current frames through the 384 x 384 resize, the look-down frame at camera size (smart_resize -> 644 x 476): the
            # rows of one sequence are [N x 784 | 1564] in prompt order
            per = self.GRID[1] * self.GRID[2]
            d3 = dst.view(m, self.pv_rows_fresh, 1176)
            pv, _ = self.pre.qwen_pixel_values(self.frames_u8[lo:lo + m, :N].reshape(m * N, 480, 640, 3))
            d3[:, : N * per].copy_(pv.view(m, N * per, 1176))
            x = self.pre.resize(self.frames_u8[lo:lo + m, N].contiguous(), self.LD_GRID[2] * 14, self.LD_GRID[1] * 14)
            pvl = torch.empty(m * self.LD_GRID[1] * self.LD_GRID[2], 1176, dtype=torch.bfloat16, device=self.dev)
            from internnav_amd import ops

            ops.qwen_patchify_u8(x, pvl, self.pre.qwen_lut, 14, 2, 2)
            d3[:, N * per:].copy_(pvl.view(m, -1, 1176))
        else:
            dst.copy_(self.pixel_values[lo:lo + m].reshape(-1, 1176))

    def _ingest_s1(self, envs=None):
        """System-1 look-down pairs (goal frame = frame 0, current = last frame) -> images_dp bf16 [B, 2, 224, 224, 3]; envs: LongTensor of
        the envs that run System-1 this step (None = all)."""
        if not self.raw:
            return
        if envs is None:
            raw = torch.index_select(self.frames_u8, 1, self.s1_sel)
            self.images_dp.copy_(self.pre.s1_frames(raw.view(self.B * 2, 480, 640, 3)).view(self.B, 2, 224, 224, 3))
        else:
            raw = torch.index_select(torch.index_select(self.frames_u8, 0, envs), 1, self.s1_sel)
            self.images_dp.index_copy_(0, envs, self.pre.s1_frames(raw.view(-1, 480, 640, 3)).view(-1, 2, 224, 224, 3))

    def _s2_call(self, m):
        s = self.s2[m]
        self.model.qwen.run_s2(s["P"], s["pv"], s["toks"], s["lat"])

    def _s1_call(self, n=None):
        n = self.B if n is None else n
        return self.model.s1.generate_traj(self.latent_table[:n], self.images_dp[:n], self.x_init[:n])

    def capture(self):
        from internnav_amd import runtime

        for m, s in self.s2.items():
            self._ingest_s2(0, m, s["pv"])
            s["graph"] = runtime.GraphedCall(lambda m=m: self._s2_call(m), {})
        if self.with_s1 and self.cadence == "nominal":
            self.s1_graph = runtime.GraphedCall(lambda: self._s1_call(), {})
        if self.overlap:
            q = self.model.qwen
            for j in range(self.PERIOD):
                m, nA = self.mb[j], len(self.idxA_host[j])
                if m == 0:              # System-1-only step (--s2-every): the all-env System-1 graph captured above
                    continue
                mg = self.merge_images
                if mg and (nA, m) not in self.gAimg:
                    # one encoder pass over the look-down pairs of the side envs AND the System-2 envs; the latter's 32 memory tokens go to the small engine
                    s1, sm = self.model.s1, self.s1_small

                    def enc(nA=nA, m=m):
                        s1.encode_images(nA + m, self.imgA[:nA + m])
                        sm.z[: m * sm.Lz].view(m, sm.Lz, sm.L)[:, :32].copy_(s1.z[: (nA + m) * s1.Lz].view(nA + m, s1.Lz, s1.L)[nA:, :32])
                    self.gAimg[(nA, m)] = runtime.GraphedCall(enc, {}, workspace_slot=1)
                if nA not in self.gA:
                    self.gA[nA] = runtime.GraphedCall(lambda nA=nA: self.model.s1.generate_traj(self.latA[:nA], self.imgA[:nA], self.xA[:nA], images_encoded=mg), {}, workspace_slot=1)
                if m not in self.gB:
                    self.gB[m] = runtime.GraphedCall(lambda m=m: self.s1_small.generate_traj(self.latB[:m], self.imgB[:m], self.xB[:m], images_encoded=mg), {}, workspace_slot=2)
                    s = self.s2[m]
                    self.gP[m] = runtime.GraphedCall(lambda s=s: q.run_prefill(s["P"], s["pv"]), {})

                    def dec(s=s):
                        q.run_decode(s["P"], s["toks"], 0, None)
                        q.run_latents(s["P"], s["lat"])
                    self.gD[m] = runtime.GraphedCall(dec, {})

    def check_overlap(self):
        """same step run with both schedules from the same state and noise: the trajectories must agree (different batch splits select
        different GEMM tile kernels, so the comparison is to bf16 tolerance, not bit-exact)."""
        self.freeze_noise = True
        lat0 = self.latent_table.clone()
        ov = self.overlap
        self.overlap = False
        self.step(0)
        t1 = self.last_traj.clone()
        self.latent_table.copy_(lat0)
        self.overlap = ov
        self.step(0)
        t2 = self.last_traj.clone()
        self.latent_table.copy_(lat0)
        self.freeze_noise = False
        self.queue = [[] for _ in range(self.B)]
        return float((t1 - t2).abs().max().item()), float(t1.abs().max().item())

    # ---- host post-processing (vln_utils.traj_to_actions per env, as internvla_n1_policy.py:205-213) and the action table of the step
    def _plan(self, traj_host):
        return [x for x in self.traj_to_actions(traj_host) if x != 0][:4]

    def _emit(self, acts, plans):
        """plans: {env: action list of its fresh System-1 plan}. nominal cadence: the row of an env = its plan (<= 4 ids, zero padded).
        reference cadence: an env executes ONE action per step - the head of its fresh plan or the next queued action of its current
        one (internvla_n1_agent.py:279-300,331-340) - reported in column 0."""
        if self.cadence == "reference":
            for b, al in plans.items():
                self.queue[b] = list(al)
            for b in range(self.B):
                acts[b, 0] = self.queue[b].pop(0) if self.queue[b] else 0
        else:
            for b, al in plans.items():
                acts[b, :len(al)] = al

    def step_overlapped(self, i):
        j = i % self.PERIOD
        m, lo = self.mb[j], int(self.mb_start[j])
        if m == 0:
            return self._step_s1_only()
        idx, hostidx = self.idxA[j], self.idxA_host[j]
        nA = len(hostidx)
        s = self.s2[m]
        main = torch.cuda.current_stream()
        if not getattr(self, "freeze_noise", False):
            self.x_init.normal_(generator=self.g)
        if self.cadence == "nominal":
            self._ingest_s1()
        else:
            self._ingest_s1(torch.cat([idx, torch.arange(lo, lo + m, device=self.dev)]))
        # side stream: System-1 for the envs keeping their current plan's latents (earlier System-2 calls)
        torch.index_select(self.latent_table, 0, idx, out=self.latA[:nA])
        torch.index_select(self.images_dp, 0, idx, out=self.imgA[:nA])
        if self.merge_images:
            self.imgA[nA:nA + m].copy_(self.images_dp[lo:lo + m])
        torch.index_select(self.x_init, 0, idx, out=self.xA[:nA])
        # main stream: System-2 micro-batch (prefill alone: MFMA bound), then decode + latent queries || side-stream System-1
        s["P"]["ids"].copy_(self.ids[lo:lo + m, self.prefix_len:].reshape(-1).to(torch.int32))
        self._ingest_s2(lo, m, s["pv"])
        self.gP[m]()
        self.ev.record(main)
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.ev)
            if self.merge_images:
                self.gAimg[(nA, m)]()
                self.ev_img.record(self.side)
            trajA = self.gA[nA]()
        self.gD[m]()
        self.latent_table[lo:lo + m].copy_(s["lat"])
        self.latB[:m].copy_(s["lat"])
        if self.merge_images:
            main.wait_event(self.ev_img)         # the memory tokens of the System-2 envs (long done: the decode chain ran meanwhile)
        else:
            self.imgB[:m].copy_(self.images_dp[lo:lo + m])
        self.xB[:m].copy_(self.x_init[lo:lo + m])
        trajB = self.gB[m]()
        # host post-processing of the side-stream envs runs while the main stream is still busy with the System-2 decode passes and the
        # System-1 call of the System-2 envs
        acts = np.zeros((self.B, 4), dtype=np.int32)
        plans = {}
        with torch.cuda.stream(self.side):
            self.hostA[:nA].copy_(trajA, non_blocking=True)
        self.side.synchronize()
        for k, b in enumerate(hostidx):
            plans[b] = self._plan(self.hostA[k])
        self.hostB[:m].copy_(trajB, non_blocking=True)
        main.synchronize()
        for k in range(m):
            plans[lo + k] = self._plan(self.hostB[k])
        self._emit(acts, plans)
        if getattr(self, "freeze_noise", False):   # schedule check: keep the assembled trajectories
            self.traj.index_copy_(0, idx, trajA)
            self.traj[lo:lo + m].copy_(trajB)
            self.last_traj = self.traj.clone()
        self.actions.copy_(torch.from_numpy(acts))
        return self.actions

    def _step_s1_only(self):
        """a step without a System-2 micro-batch (--s2-every E > 1): System-1 for every env on the latents their last System-2 call left, one
        graph replay; the host post-processing of the first half of the envs runs while the second half's trajectories are still in flight."""
        if not getattr(self, "freeze_noise", False):
            self.x_init.normal_(generator=self.g)
        self._ingest_s1()
        traj = self.s1_graph() if self.s1_graph else self._s1_call()
        self.last_traj = traj
        t = traj.cpu()
        acts = np.zeros((self.B, 4), dtype=np.int32)
        self._emit(acts, {b: self._plan(t[b]) for b in range(self.B)})
        self.actions.copy_(torch.from_numpy(acts))
        return self.actions

    def step(self, i):
        if self.overlap:
            return self.step_overlapped(i)
        j = i % self.PERIOD
        m, lo = self.mb[j], int(self.mb_start[j])
        if m == 0:
            return self._step_s1_only()
        s = self.s2[m]
        # System-2 for the envs whose plan expires this step: gather their prompt / frames, run, scatter the latents back
        s["P"]["ids"].copy_(self.ids[lo:lo + m, self.prefix_len:].reshape(-1).to(torch.int32))
        self._ingest_s2(lo, m, s["pv"])
        s["graph"]() if s["graph"] else self._s2_call(m)
        self.latent_table[lo:lo + m].copy_(s["lat"])
        if not self.with_s1:
            self.actions[lo:lo + m, 0].copy_(s["toks"][:, 0] % 4)           # something that depends on the call's output
            return self.actions
        if not getattr(self, "freeze_noise", False):
            self.x_init.normal_(generator=self.g)
        acts = np.zeros((self.B, 4), dtype=np.int32)
        if self.cadence == "nominal":
            # System-1 for every env
            self._ingest_s1()
            traj = self.s1_graph() if self.s1_graph else self._s1_call()
            self.last_traj = traj
            t = traj.cpu()                                        # [B, 32, 32, 3] -> host post-processing of the reference (vln_utils)
            self._emit(acts, {b: self._plan(t[b]) for b in range(self.B)})
        else:
            # reference cadence on one stream: System-1 for the System-2 envs and the envs half a period away, eager gather / scatter
            envs = self.idxA_host[j] + list(range(lo, lo + m))
            e = torch.tensor(envs, device=self.dev)
            self._ingest_s1(e)
            n = len(envs)
            traj = self.model.s1.generate_traj(self.latent_table[e].contiguous(), self.images_dp[e].contiguous(), self.x_init[e].contiguous())
            full = torch.zeros(self.B, *traj.shape[1:], device=self.dev)
            full[e] = traj
            self.last_traj = full
            t = traj.cpu()
            self._emit(acts, {b: self._plan(t[k]) for k, b in enumerate(envs)})
            assert n <= self.model.s1.b_max
        self.actions.copy_(torch.from_numpy(acts))
        return self.actions

    def step_output_for_check(self):
        return self.actions

    def instrumented(self):
        m = max(self.mb)
        q = self.model.qwen
        # the launches of the timed step (two half micro-batches, tiles selected in the shared-tail mode), issued back to back on ONE stream
        # so that the per-launch HIP events time kernels that do not overlap (round 3 timed the joint launch sequence instead: a tile
        # selection the timed step does not run)
        keep, q.split_serial = q.split_serial, True
        try:
            self._s2_call(m)
        finally:
            q.split_serial = keep
        if self.with_s1:
            n = self.B if self.cadence == "nominal" else min(self.model.s1.b_max, 2 * m)
            self._s1_call(n)
            return f"one System-1 call over {n} envs + one System-2 call over {m} envs"
        return f"one System-2 call over {m} envs"

    def cpu_baseline(self):
        return n1_cpu_baseline(self.qcfg, self.grids_seq, self.pv_rows_seq, self.ids[:1].cpu().long(), self.N_DECODE, self.cadence, self.with_s1, self.unit,
                               scfg=self.scfg)


def n1_cpu_baseline(qc, grids, pv_rows, ids, n_decode, cadence, with_s1, unit, depth_cycle: int = 2, scfg=None):
    """the reference PyTorch path on the host cores = the CPU oracle, ONE env, bf16 (weights held in bf16 as the reference loads them,
    torch.autocast for the activations), at FULL depth and AS THE REFERENCE EXECUTES a pixel-goal System-2 call
    (internvla_n1_policy.py:163-199): ViT (v_depth blocks) + prefill (t_layers layers, lm_head on every position, internvla_n1.py:220) +
    n_decode - 1 cached single-token steps of `generate`, then `generate_latents` = a SECOND ViT + prefill over the prompt + answer + the TRAJ
    tokens (internvla_n1.py:320-347); and the complete System-1 call. System-2 is timed once at full depth (the layer executions cycle
    through `depth_cycle` distinct sets of layer weights - the timing does not depend on the values, and drawing 15 GB of weights does not
    fit a bounded sample), System-1 as 1 warm-up + the median of 3 (fp32) / 2 (bf16) runs, the faster counts. The round-3 sample (two
    blocks + two layers scaled linearly) is kept as a cross-check. qc: the Qwen configuration; grids: the images of one prompt; ids [1, S]."""
    from internnav_amd import synthetic
    from oracle import nextdit as o_nd  # cpu_baseline leg only
    from oracle import qwen_vl as o_q

    cores = min(64, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    out = {}
    t_s1 = 0.0
    if with_s1:
        sd = synthetic.n1_nextdit_state_dict(0, cfg=scfg or synthetic.N1_NEXTDIT_CFG)
        inp = synthetic.n1_nextdit_inputs(1, 0)
        s1 = lambda: o_nd.generate_traj(sd, inp["traj_latents"], inp["images"], inp["x_init"])  # noqa: E731
        t32, t16 = _median_time(s1, 3), _median_time(s1, 2, autocast=True)
        t_s1 = min(t32, t16)
        out["s1_call_s"] = {"fp32": round(t32, 2), "bf16_autocast": round(t16, 2)}
    # ---- System-2 at full depth: `depth_cycle` sets of layer weights at the true widths, bf16; embed_tokens aliases lm_head (same shape)
    cfg2 = dict(qc, v_depth=depth_cycle, v_fullatt=(depth_cycle - 1,), t_layers=depth_cycle)
    spec = dict(synthetic.qwen_spec(cfg2))
    spec.pop("model.embed_tokens.weight")
    sd2 = {k: (v.to(torch.bfloat16) if v.dim() >= 2 else v) for k, v in synthetic.materialize(spec, 0).items()}
    sd2["model.embed_tokens.weight"] = sd2["lm_head.weight"]

    class Cyclic(dict):
        """state dict of the FULL configuration whose block / layer i reads the weights of block / layer i % depth_cycle"""

        def __missing__(self, key):
            for pre in ("visual.blocks.", "model.layers."):
                if key.startswith(pre):
                    i, rest = key[len(pre):].split(".", 1)
                    return self[f"{pre}{int(i) % depth_cycle}.{rest}"]
            raise KeyError(key)
    sdc = Cyclic(sd2)
    S = ids.shape[1]
    pv = torch.randn(pv_rows, 1176).to(torch.bfloat16)
    t = {}

    def timed(name, fn):
        t0 = time.time()
        r = fn()
        t[name] = time.time() - t0
        return r

    def s2_as_executed():
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            img = timed("vit", lambda: o_q.vision_tower(pv, grids, sdc, qc))
            x = o_q.input_embeds(ids, img, sdc, qc)
            pos, _ = o_q.rope_index(ids, grids, qc["image_token_id"], qc["vision_start_id"])
            cache = [None] * qc["t_layers"]
            h = timed("prefill", lambda: o_q.decoder_stack(x, pos, sdc, qc, cache=cache))
            logits = timed("lm_head_all_positions", lambda: torch.nn.functional.linear(h, sdc["lm_head.weight"]))
            toks = [logits[:, -1].argmax(-1)]
            t0 = time.time()
            for jj in range(n_decode - 1):
                xe = sdc["model.embed_tokens.weight"][toks[-1]][:, None]
                h1 = o_q.decoder_stack(xe, pos[:, :, -1:] + 1 + jj, sdc, qc, cache=cache)
                toks.append(torch.nn.functional.linear(h1, sdc["lm_head.weight"])[:, -1].argmax(-1))
            t["decode_cached_steps"] = time.time() - t0
            # generate_latents: the whole forward again on prompt + answer + N_QUERY TRAJ tokens
            ids2 = torch.cat([ids, torch.stack(toks, 1), torch.full((1, qc["n_query"]), qc["traj_token_id"], dtype=ids.dtype)], 1)
            img2 = timed("latents_vit", lambda: o_q.vision_tower(pv, grids, sdc, qc))
            x2 = o_q.input_embeds(ids2, img2, sdc, qc)
            pos2, _ = o_q.rope_index(ids2, grids, qc["image_token_id"], qc["vision_start_id"])
            timed("latents_prefill", lambda: o_q.decoder_stack(x2, pos2, sdc, qc))

    # warm-up at `depth_cycle` layers on a short input (pages the kernels in), then ONE timed run at full depth
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        n0 = grids[0][0] * grids[0][1] * grids[0][2]
        o_q.vision_tower(pv[:n0], grids[:1], sd2, cfg2)
        o_q.decoder_stack(torch.randn(1, 64, qc["t_hidden"]), torch.arange(64).view(1, 1, -1).expand(3, 1, -1), sd2, cfg2)
    s2_as_executed()
    t_exec = sum(t.values())
    t_once = t["vit"] + t["prefill"] + t["decode_cached_steps"]      # + the latent queries on the cache (4 rows): less than one decode step
    out["s2_call_s"] = {k: round(v, 2) for k, v in t.items()}
    out["s2_call_as_executed_s"] = round(t_exec, 2)
    out["s2_call_without_the_second_forward_s"] = round(t_once, 2)
    # cross-check: the round-3 sample
    xs = torch.randn(1, S, qc["t_hidden"])
    ps = torch.arange(S).view(1, 1, -1).expand(3, 1, -1)
    c_vit = _median_time(lambda: o_q.vision_tower(pv, grids, sd2, cfg2), 2, autocast=True) * qc["v_depth"] / depth_cycle
    c_llm = _median_time(lambda: o_q.decoder_stack(xs, ps, sd2, cfg2), 2, autocast=True) * qc["t_layers"] / depth_cycle
    out["cross_check_scaled_sample_s"] = {"vit": round(c_vit, 2), "prefill": round(c_llm, 2)}
    if cadence == "nominal":
        t_step, how = t_s1 + t_exec / 10, "S1 call + 0.1 x S2 call"
    elif cadence == "reference":
        t_step, how = (t_exec + 2 * t_s1) / 8, "(S2 call + 2 S1 calls) / 8"
    else:
        t_step, how = t_exec, "one S2 call"
    return dict({"value": round(1.0 / t_step, 4), "unit": unit, "cores": cores, "cpu": _cpu_model(), "kind": "port",
                 "sample": f"1 env, batch-1 as the reference executes, torch CPU bf16 (bf16 weights + autocast): full-depth System-2 call as executed by the "
                           f"reference (ViT {qc['v_depth']} blocks + prefill {qc['t_layers']} layers at S={S} + lm_head on all positions + {n_decode - 1} cached decode "
                           f"steps + generate_latents' second ViT + prefill), timed once after a short warm-up, layer weights cycling through {depth_cycle} sets"
                           + ("; System-1 call: 1 warm-up + median of 3 (fp32) / 2 (bf16) runs, the faster counts" if with_s1 else "")
                           + f"; combined as {how}"}, **out)


def n1_variants(a, steps: int = 10, warmup: int = 3) -> dict:
    """the headline workload under its two EXACT schedule variants and at the other FFN width, `steps` timed steps each on this same box, for
    the driver's record (the headline `value` is not touched): prefix_kv = K/V of template + instruction + frame 0 from a per-env cache
    (--prefix-kv: causal attention, bit-exact), s2_every2 = System-2 micro-batches of 12-14 envs on every other step (--s2-every 2: same calls per
    env, the single-token passes stream the decoder weights half as often; worse step-latency spread), dit_ffn_1024 = the NextDiT FFN of the
    diffusers <= 0.32 convention (what rounds 1-5 timed). Each variant is this same file run in a fresh process (its line is what
    `python bench.py <flag> --steps 10` prints): timed inside the headline's process the variants with eager host-side work per step came out
    8-10 % low (profiles/r06j_variants_inline_vs_process.txt)."""
    import subprocess

    out = {}
    for name, flags in (("prefix_kv", ["--prefix-kv"]), ("s2_every2", ["--s2-every", "2"]), ("dit_ffn_1024", ["--dit-ffn", "1024"])):
        cmd = [sys.executable, str(Path(__file__).resolve()), "--gpus", "1", "--steps", str(steps), "--warmup", str(warmup), "--envs", str(a.envs),
               "--no-cpu-baseline", "--no-variants"] + flags
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
            d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            out[name] = {"value": d["value"], "steps": d["steps"], "ms_per_step": d["ms_per_step"], "p50": d["step_latency_ms"]["p50"], "max": d["step_latency_ms"]["max"],
                         "workload": d["config"]["workload"], "algorithmic_tflop_per_env_step": d["roofline"]["whole_step"]["algorithmic_tflop_per_env_step"],
                         "cmd": "python bench.py " + " ".join(cmd[2:])}
        except Exception as e:      # a variant must never cost the headline line
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def _median_time(fn, runs: int, autocast: bool = False) -> float:
    """wall-clock of fn(): one untimed warm-up, then the median of `runs` runs (>= 3 runs after warm-up)."""
    ts = []
    with torch.no_grad(), torch.autocast(device_type="cpu", dtype=torch.bfloat16, enabled=autocast):
        fn()
        for _ in range(runs):
            t0 = time.time()
            fn()
            ts.append(time.time() - t0)
    return sorted(ts)[len(ts) // 2]


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def pmc_traffic(workload, kernel=None):
    """HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/ (rocprofv3 cannot run inside the
    bench): average FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE per GEMM launch of the same workload, or None if not collected.
    Entries may carry `by_kernel` (bench-line kernel name -> entry): the one of the kernel the line names is reported."""
    f = ROOT / "profiles" / "pmc_traffic.json"
    if not f.exists():
        return None
    t = json.loads(f.read_text()).get(workload)
    if not t:
        return None
    if "by_kernel" in t:
        t = t["by_kernel"].get(kernel)
        if not t:
            return None
    return {"bytes_per_launch": round((t["read_MB_per_launch_x2_corrected"] + t["write_MB_per_launch"]) * 1e6), "kernel": t["kernel"], "source": t["source"]}


# ------------------------------------------------------------------------------------------------------------ driver
def main():
    a = parse()
    from internnav_amd.dist import maybe_self_spawn

    maybe_self_spawn(str(Path(__file__).resolve()), a.gpus)      # `python bench.py --gpus N` without a launcher starts its own N ranks
    if a.workload == "sft":                                       # config #5 in the same JSON shape (bench_sft.py)
        import bench_sft

        sys.argv = [str(ROOT / "bench_sft.py"), "--gpus", str(a.gpus), "--steps", str(a.steps), "--warmup", str(a.warmup)] + \
                   (["--no-cpu-baseline"] if a.no_cpu_baseline else []) + a.rest
        return bench_sft.main()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from internnav_amd.dist import under_launcher

    # a scheduler's WORLD_SIZE without a rank is not a launcher: only a launcher's environment makes this process one of N ranks
    world = int(os.environ["WORLD_SIZE"]) if under_launcher() else 1
    host_stub = a.workload == "host_stub"
    from internnav_amd import runtime

    if host_stub:
        dev, arch = torch.device("cpu"), "cpu (host stub, gloo)"
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        arch = runtime.require_gfx950()
    dist = None
    pinned = None
    if world > 1:
        import torch.distributed as dist

        from internnav_amd.dist import pin_host_threads

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if host_stub:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
        pinned = pin_host_threads(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))   # own slice of the host cores per rank
    t_setup = time.perf_counter()
    wl = {"n1_dual": N1Dual, "s2_only": N1Dual, "navdp_s1": NavDPS1, "unet1d_s1": UNet1DS1, "host_stub": HostStub}[a.workload](a, dev, rank)
    t_weights = time.perf_counter() - t_setup
    if not a.no_graph:
        wl.capture()
    if not host_stub:
        torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    # what the process group REALLY is (not an echo of the environment): backend and size as torch.distributed reports them, and every
    # rank's setup time (weights + graph capture) so an N-rank run is known to fit the driver's timeout
    dist_info = {"backend": None, "ranks": 1, "setup_s_per_rank": [round(t_setup, 1)], "weights_s_per_rank": [round(t_weights, 1)]}
    if world > 1:
        ts = torch.tensor([t_setup, t_weights], device=dev, dtype=torch.float64)
        allt = torch.empty(world * 2, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(allt, ts)
        allt = allt.view(world, 2).cpu()
        dist_info = {"backend": str(dist.get_backend()), "ranks": int(dist.get_world_size()),
                     "setup_s_per_rank": [round(float(x), 1) for x in allt[:, 0]], "weights_s_per_rank": [round(float(x), 1) for x in allt[:, 1]]}
    gathered = torch.empty((world * wl.action_shape[0],) + tuple(wl.action_shape[1:]), device=dev,
                           dtype=torch.int32 if a.workload in ("n1_dual", "s2_only", "host_stub") else torch.float32) if world > 1 else None

    def step(i):
        out = wl.step(i)
        if world > 1:
            dist.all_gather_into_tensor(gathered, out)  # per-env action outputs to every rank (RCCL over xGMI)

    def sync():
        if not host_stub:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            if not host_stub:
                torch.cuda.synchronize()

    overlap_check = None
    if getattr(wl, "overlap", False):
        overlap_check = wl.check_overlap()
    for i in range(a.warmup):
        step(i)
    sync()
    t0 = time.perf_counter()
    marks = [t0]
    for i in range(a.steps):
        step(a.warmup + i)
        marks.append(time.perf_counter())          # (a step returns after its host post-processing: its actions exist at this point)
    sync()
    dt = time.perf_counter() - t0
    lat = np.diff(np.asarray(marks)) * 1e3
    step_latency = {"p50": round(float(np.percentile(lat, 50)), 2), "min": round(float(lat.min()), 2), "max": round(float(lat.max()), 2)}
    if world > 1:
        # the exchanged actions are really everybody's: this rank's slice equals its own last output, every slice is a valid action table
        mine = wl.step_output_for_check()
        assert torch.equal(gathered[rank * wl.B:(rank + 1) * wl.B], mine), "all_gather: own slice differs from the local actions"
        if a.workload in ("n1_dual", "s2_only", "host_stub"):
            assert int(gathered.min()) >= 0 and int(gathered.max()) <= 3, "all_gather: action ids outside {0..3}"
        chk = torch.stack([gathered[r * wl.B:(r + 1) * wl.B].double().sum() for r in range(world)])
        ref = chk.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(chk, ref), "all_gather: ranks hold different gathered tensors"
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    value = world * wl.B * a.steps / dt

    if rank == 0 and host_stub:
        # rank 0 alone in the post-run section (the GPU workloads run their instrumented pass and the CPU baseline here) while the others wait
        time.sleep(0.2)
        print(json.dumps({"metric": "policy steps/sec/node", "value": round(value, 2), "unit": "policy steps/s", "n_gpus": world, "steps": a.steps,
                          "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "int32", "data": "synthetic trajectories", "rccl_ranks": 0, "gloo_ranks": dist_info["ranks"] if dist_info["backend"] == "gloo" else 0,
                          "dist": dist_info,
                          "config": dict({"workload": wl.name, "envs_per_gpu": wl.B, "parallelism": f"dp{world}", "device": arch,
                                          "host_cores_per_rank": pinned, "host_cores": len(os.sched_getaffinity(0)) if pinned is None else None}, **wl.desc),
                          "roofline": None, "cpu_baseline": None}), flush=True)
    elif rank == 0:
        calib = calibration_gemm(dev)
        # ---- roofline: one instrumented eager pass, HIP events around every launch on the launch stream
        runtime.prof_enable(True)
        extra = wl.instrumented()
        torch.cuda.synchronize()
        prof = runtime.prof_read()
        per_kernel = runtime.prof_read_gemm_kernels()
        runtime.prof_enable(False)               # (also clears the tally)
        gm = prof["gemm"]
        blend = gm["flops"] / (gm["ms"] * 1e-3) / 1e12 if gm["ms"] > 0 else 0.0
        # the dominant kernel = the named kernel with the largest share of the step's GEMM time
        dom_name, dom = max(per_kernel.items(), key=lambda kv: kv[1]["ms"]) if per_kernel else ("none", dict(ms=0.0, launches=0, flops=0.0))
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
        step_tflops = (value / world) * wl.f_alg / 1e12
        roofline = {
            "bound": "mfma", "kernel": dom_name,
            "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
            # the same fraction over ALL tiled MFMA GEMM launches of the pass: a launch moving to another tile kernel cannot change it
            "frac_gemm_class_blend": round(blend / PEAK_BF16_TFLOPS, 4),
            "launches": dom["launches"], "avg_launch_us": round(dom["ms"] * 1e3 / max(dom["launches"], 1), 2),
            "algorithmic_tflop_per_launch": round(dom["flops"] / max(dom["launches"], 1) / 1e12, 4),
            "algorithmic_mbytes_per_launch": round(dom.get("bytes", 0.0) / max(dom["launches"], 1) / 1e6, 1),      # operands + output once (2MK + 2NK + out)
            "traffic": pmc_traffic(a.workload, dom_name),
            "gemm_class_blend": {"what": "all tiled MFMA GEMM launches of the instrumented pass (every tile config)", "achieved": round(blend, 1),
                                 "frac": round(blend / PEAK_BF16_TFLOPS, 4)},
            "per_kernel": {k: {"launches": v["launches"], "ms": round(v["ms"], 3), "tflop": round(v["flops"] / 1e12, 3), "gbyte": round(v.get("bytes", 0.0) / 1e9, 2),
                               "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1)} for k, v in per_kernel.items()},
            "instrumented_pass": {"what": extra if isinstance(extra, str) else "one engine call over all envs",
                                  "gemm_launches": gm["launches"], "gemm_avg_launch_us": round(gm["ms"] * 1e3 / max(gm["launches"], 1), 2),
                                  "gemm_tflop": round(gm["flops"] / 1e12, 3),
                                  "kernel_class_ms": {k: round(v["ms"], 3) for k, v in prof.items()},
                                  "kernel_class_tflops": {k: round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1) for k, v in prof.items()},
                                  "kernel_class_algorithmic_GBps": {k: round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1) for k, v in prof.items()}},
            "whole_step": {"algorithmic_tflop_per_env_step": round(wl.f_alg / 1e12, 4),
                           "achieved_tflops_per_gpu": round(step_tflops, 1), "frac": round(step_tflops / PEAK_BF16_TFLOPS, 4)},
        }
        line = {
            "metric": "policy steps/sec/node" if getattr(wl, "unit", "policy steps/s") == "policy steps/s" else "System-2 calls/sec/node",
            "value": round(value, 2), "unit": getattr(wl, "unit", "policy steps/s"), "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "step_latency_ms": step_latency, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic (seeded random weights at the true shapes, synthetic camera frames / prompts)",
            # ranks of the process group torch.distributed initialised on backend "nccl" (= RCCL on ROCm); 1 = no process group (single GPU)
            "rccl_ranks": dist_info["ranks"] if (world == 1 or dist_info["backend"] == "nccl") else 0, "dist": dist_info,
            "config": dict({"workload": wl.name, "envs_per_gpu": wl.B, "parallelism": f"dp{world}", "calibration": calib,
                            "launch": "eager" if a.no_graph else "hipGraph replay",
                            "schedule": ("S2 ViT+prefill, then S2 decode+latent queries || S1(envs keeping their latents) on a side stream, then S1(S2 envs)"
                                         if getattr(wl, "overlap", False) else "single stream"),
                            "device": arch}, **wl.desc),
            "roofline": roofline,
        }
        if overlap_check is not None:
            line["config"]["schedule_check"] = {"max_abs_diff_vs_single_stream": round(overlap_check[0], 5), "traj_abs_max": round(overlap_check[1], 3)}
        plain = (a.workload == "n1_dual" and a.cadence == "nominal" and not (a.prefix_kv or a.vit_cache or a.lookdown or a.no_graph or a.no_overlap) and
                 a.s2_every == 1 and a.num_history == 3 and a.dit_ffn == 1536)
        cpu_leg = wl.cpu_baseline if (world == 1 and not a.no_cpu_baseline) else None
        if world == 1 and plain and not a.no_variants:
            # (before the CPU leg: that one leaves 64 busy host threads behind it, and the variants' steps carry host work)
            line["variants"] = n1_variants(a)
        # the CPU port of the reference path is timed on rank 0 of a single-GPU run only (contract); the key is always present
        nthr = torch.get_num_threads()
        line["cpu_baseline"] = cpu_leg() if cpu_leg is not None else None
        torch.set_num_threads(nthr)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
