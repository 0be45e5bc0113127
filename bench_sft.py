#!/usr/bin/env python
"""Benchmark of the SFT step (BASELINE config #5: InternVLA-N1 bf16 SFT step, fused AdamW, frozen System-2): samples / sec / node.

  python bench_sft.py [--gpus N] [--steps K] [--warmup W] [--micro-batch 2] [--subgoals 12] [--frames 10]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench_sft.py --gpus N ...

`bench.py` stays the headline (policy steps/s); this is the measurement of SURVEY.md 8 row f4 in the same JSON shape.
One step = one micro-batch per rank through `InternVLAN1SftTrainer.training_step`: frozen Qwen2.5-VL-7B ViT + prefill of `frames`
images (392x392 -> 196 tokens each) + text per sample, the latent-query rows, NextDiT-async System-1 loss / backward over
micro_batch x subgoals (sample, sub-goal) pairs, the latent-query backward through the frozen LLM, the gradient all-reduce of the
ONE flat fp32 bucket over RCCL (N > 1) and the fused clip + AdamW launch. Weights are seeded random at the true shapes, inputs
synthetic and resident in HBM. The reference trains with per-device batch 2 (train_dual_system.sh:23); global batch = 2 x N x
accumulation, which does not change the per-sample cost measured here ("scaling": "weak").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
PEAK_BF16_TFLOPS = 2500.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--micro-batch", type=int, default=2, help="samples per rank per step (reference: per_device_train_batch_size 2)")
    ap.add_argument("--subgoals", type=int, default=12, help="sub-goal frames per sample (dataset max_len = 12)")
    ap.add_argument("--frames", type=int, default=10, help="images in the System-2 prompt (8 history + current + look-down)")
    ap.add_argument("--zero2", action="store_true", help="reduce-scatter + sharded update + all-gather instead of all-reduce")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="frozen prefix (ViT + prefill) of every micro-batch inside its own step, on the main stream (round-3 / early round-4 schedule) "
                         "instead of one step ahead on the prefetch stream (InternVLAN1SftTrainer.prefetch)")
    ap.add_argument("--prefetch-first", action="store_true", help="A/B: issue the next prefix before this step's own launches (first version of the pipeline)")
    ap.add_argument("--no-graph-prefix", action="store_true",
                    help="A/B: the frozen prefix (ViT + ragged prefill, ~1 100 launches) as eager launches instead of ONE hipGraph replay per prompt geometry. "
                         "Eager, the step depends on the box's host (14.3-17.7 samples/s over boxes and runs); graphed AND with the step's own launches on a "
                         "high-priority stream it is 17.6-17.9 on every box (profiles/r06z_sft_*_ab.txt)")
    ap.add_argument("--no-priority", action="store_true", help="A/B: the step's own launches on the caller's stream instead of a high-priority one")
    ap.add_argument("--no-split-prefill", action="store_true", help="A/B: the frozen prefix as one launch sequence instead of two half micro-batches on two streams")
    ap.add_argument("--no-graph-s1", action="store_true", help="System-1 loss + backward as eager launches (round-3 path) instead of one hipGraph replay")
    return ap.parse_args()


def build(a, dev, rank):
    from internnav_amd import synthetic
    from internnav_amd.qwen_vl import QwenVLEngine
    from internnav_amd.trainer import InternVLAN1SftTrainer

    qcfg = synthetic.QWEN_N1_CFG
    B, T, F = a.micro_batch, a.subgoals, a.frames
    per = 784
    n_text, n_tail = 98, 30
    S = n_text + F * (per // 4 + 2) + n_tail
    weights = synthetic.LazyDeviceWeights(synthetic.qwen_spec(qcfg), dev, seed=0)
    eng = QwenVLEngine(weights, qcfg, dev, max_seqs=B, max_seq_len=(S + qcfg["n_query"] + 63) // 64 * 64, max_patches=B * F * per)
    if getattr(a, "no_split_prefill", False):
        eng.split_prefill = False
    sd_s = {k: v.float() for k, v in synthetic.materialize(synthetic.n1_nextdit_spec(), 0).items()}
    from internnav_amd.dist import under_launcher

    world = int(os.environ["WORLD_SIZE"]) if under_launcher() else 1
    tr = InternVLAN1SftTrainer(eng, sd_s, dev, total_steps=1000, zero2=a.zero2, graph_s1=not a.no_graph_s1, graph_prefix=not getattr(a, "no_graph_prefix", False))
    tr.step_idx = 10          # past the warm-up: non-zero learning rate
    tr.prefetch_first = bool(getattr(a, "prefetch_first", False))
    tr.priority_step = not getattr(a, "no_priority", False)
    g = torch.Generator(device=dev).manual_seed(1000 * rank + 7)
    lim = qcfg["image_token_id"] - 16
    ids = torch.randint(0, lim, (B, S + qcfg["n_query"]), device=dev, generator=g)
    o = n_text
    for _ in range(F):
        ids[:, o] = qcfg["vision_start_id"]
        ids[:, o + 1:o + 1 + per // 4] = qcfg["image_token_id"]
        ids[:, o + 1 + per // 4] = qcfg["vision_end_id"]
        o += per // 4 + 2
    ids[:, S:] = qcfg["traj_token_id"]
    batch = dict(input_ids=ids.cpu(), t_s_pos=[S] * B,
                 pixel_values=torch.randn(B * F * per, 1176, device=dev, generator=g).to(torch.bfloat16),
                 image_grid_thw=torch.tensor([[1, 28, 28]] * (B * F)),
                 traj_images=torch.rand(B, T, 224, 224, 3, device=dev, generator=g),
                 traj_poses=torch.randn(B, T, 32, 3, device=dev, generator=g).cpu(),
                 video_frame_num=torch.tensor([T] * B))
    return tr, batch, dict(S=S, B=B, T=T, F=F, world=world), qcfg


def flops_per_step(qcfg, S, B, T, F):
    """algorithmic FLOPs of one step (per rank): frozen S2 forward + 3x the System-1 forward (forward + dX + dW) + the latent-query rows."""
    from internnav_amd import flops, synthetic

    f2 = flops.s2_call_flops(S, [(1, 28, 28)] * F, 0, qcfg)["total"] * B
    scfg = synthetic.N1_NEXTDIT_CFG
    N = B * T
    s1 = flops.nextdit_s1_flops_per_env(scfg)
    dino = flops.vit_s_flops() * N                           # one ViT-S pass per distinct frame (the goal frame is frame 0 of its sample)
    per_pair = s1["memory_encoder"] + s1["qformer"] + s1["cond"] + s1["dit"] / (scfg["num_inference_steps"] * scfg["sample_num"])   # ONE DiT pass on ONE trajectory
    fwd_s1 = dino + per_pair * N
    rows = B * qcfg["n_query"]
    H, TI, L = qcfg["t_hidden"], qcfg["t_inter"], qcfg["t_layers"]
    llm_rows = 2.0 * rows * L * (H * (H + 2 * qcfg["t_kv_heads"] * (H // qcfg["t_heads"])) + H * H + 3 * H * TI) * 2    # forward + dX
    return dict(s2_forward=f2, s1_forward=fwd_s1, s1_total=3 * fwd_s1, llm_rows=llm_rows, total=f2 + 3 * fwd_s1 + llm_rows)


def cpu_baseline(info, qcfg):
    """bounded CPU sample of the same step with torch autograd on the oracle (fp32 and bf16 autocast, the faster is reported)."""
    from internnav_amd import synthetic
    from oracle import qwen_vl as o_q  # cpu_baseline leg only
    from oracle import sft as o_sft

    cores = min(64, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    sd = {k: v.float().requires_grad_(True) for k, v in synthetic.materialize(synthetic.n1_nextdit_spec(), 0).items()}
    g = torch.Generator().manual_seed(0)
    Tn = 2
    inp = dict(hq=torch.randn(1, 4, 3584, generator=g), img=torch.rand(1, Tn, 224, 224, 3, generator=g), poses=torch.randn(1, Tn, 32, 3, generator=g),
               noise=torch.randn(Tn, 32, 3, generator=g), ti=torch.randint(0, 1000, (Tn,), generator=g))
    cfg = dict(qcfg, v_depth=2, v_fullatt=(1,), t_layers=2, vocab=8)
    sdq = synthetic.materialize({k: v for k, v in synthetic.qwen_spec(cfg).items()}, 0)
    pv = torch.randn(info["F"] * 784, 1176)
    x = torch.randn(1, info["S"], qcfg["t_hidden"])
    pos = torch.arange(info["S"]).view(1, 1, -1).expand(3, 1, -1)

    def s1():
        for p in sd.values():
            p.grad = None
        o_sft.nextdit_sft_loss(sd, inp["hq"], inp["img"], inp["poses"], torch.tensor([Tn]), inp["noise"], inp["ti"]).backward()

    def timed(fn, runs, autocast, grad):
        ts = []
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast), torch.set_grad_enabled(grad):
            fn()
            for _ in range(runs):
                t0 = time.time()
                fn()
                ts.append(time.time() - t0)
        return sorted(ts)[len(ts) // 2]

    out = {}
    for name, ac in (("fp32", False), ("bf16_autocast", True)):
        t_s1 = timed(s1, 2, ac, True) * info["T"] / Tn
        t_vit = timed(lambda: o_q.vision_tower(pv, [(1, 28, 28)] * info["F"], sdq, cfg), 2, ac, False) * qcfg["v_depth"] / 2
        t_llm = timed(lambda: o_q.decoder_stack(x, pos, sdq, cfg), 2, ac, False) * qcfg["t_layers"] / 2
        out[name] = {"samples_per_s": round(1.0 / (t_s1 + t_vit + t_llm), 5),
                     "seconds_per_sample": {"s1_fwd_bwd_scaled": round(t_s1, 2), "s2_vit_scaled": round(t_vit, 2), "s2_prefill_scaled": round(t_llm, 2)}}
    best = max(out.values(), key=lambda v: v["samples_per_s"])
    cpu = "unknown"
    try:
        cpu = next(line.split(":", 1)[1].strip() for line in open("/proc/cpuinfo") if line.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return {"value": best["samples_per_s"], "unit": "samples/s", "cores": cores, "cpu": cpu, "kind": "port", **out,
            "sample": f"1 sample: System-1 loss + autograd on 2 of {info['T']} sub-goals (scaled linearly) + frozen S2 forward on 2 of 32 ViT blocks / 2 of 28 "
                      f"decoder layers at {info['F']} frames, S={info['S']} (scaled linearly); torch CPU, 1 warm-up + median of 2 runs per leg; the "
                      "latent-query backward and the optimiser step are not in the CPU sample (< 2 % of the work)"}


def main():
    a = parse()
    from internnav_amd.dist import maybe_self_spawn

    maybe_self_spawn(str(Path(__file__).resolve()), a.gpus)
    from internnav_amd.dist import under_launcher

    # a scheduler's WORLD_SIZE without a rank is not a launcher
    rank, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ["WORLD_SIZE"]) if under_launcher() else 1
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from internnav_amd import runtime

    arch = runtime.require_gfx950()
    dist = None
    if world > 1:
        import torch.distributed as dist

        from internnav_amd.dist import pin_host_threads

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        pin_host_threads(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    tr, batch, info, qcfg = build(a, dev, rank)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    losses = []
    # two batch objects stand for consecutive micro-batches of a data loader (same synthetic content): with the prefetch pipeline the frozen
    # prefix of micro-batch i + 1 runs on a second stream / engine twin while step i's System-1 loss, backward and optimiser launch run -
    # every step still does one prefix, one loss / backward and one update
    batches = [batch, dict(batch)]
    pipe = not a.no_prefetch

    def one(i):
        return tr.training_step(batches[i % 2], next_batch=batches[(i + 1) % 2] if pipe else None)
    for i in range(a.warmup):
        one(i)
    sync()
    t0 = time.perf_counter()
    for i in range(a.warmup, a.warmup + a.steps):
        losses.append(one(i))
    sync()
    dt = time.perf_counter() - t0
    tr.drop_prefetch()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # data-parallel consistency: every rank holds the same weights after the same reduced updates
        chk = tr.P.p32.double().sum().view(1)
        ref = chk.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(chk, ref), "ranks diverged: the reduced gradients / updates differ"
    value = world * info["B"] * a.steps / dt
    if rank == 0:
        fl = flops_per_step(qcfg, info["S"], info["B"], info["T"], info["F"])
        runtime.prof_enable(True)
        tr.graph_prefix = False           # the instrumented pass times every launch of the prefix with HIP events: eager
        tr.forward_backward(batch)
        torch.cuda.synchronize()
        prof = runtime.prof_read()
        per_kernel = runtime.prof_read_gemm_kernels()
        runtime.prof_enable(False)
        tr.P.zero_grad()
        dom_name, dom = max(per_kernel.items(), key=lambda kv: kv[1]["ms"]) if per_kernel else ("none", dict(ms=0.0, launches=0, flops=0.0))
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
        step_tflops = fl["total"] * a.steps / dt / 1e12
        line = {
            "metric": "SFT samples/sec/node", "value": round(value, 3), "unit": "samples/s", "n_gpus": world,
            "rccl_ranks": int(dist.get_world_size()) if dist is not None else 1, "dist_backend": str(dist.get_backend()) if dist is not None else None, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic (seeded random weights at the true shapes, synthetic prompts / frames / trajectories)",
            "config": {"workload": f"sft_nextdit_async_b{info['B']}x{info['T']}", "micro_batch_per_gpu": info["B"], "subgoals_per_sample": info["T"],
                       "s2_prompt": f"{info['F']} frames x 784 patches + text, S={info['S']} + 4 <traj> tokens", "parallelism": f"dp{world}" + ("-zero2" if a.zero2 else ""),
                       "trainable_parameters": int(sum(int(np.prod(s)) for _, s in tr.P.index.values())), "optimizer": "fused AdamW + clip 1.0, cosine_with_min_lr", "dropout": 0.1,
                       "launch": ("frozen prefix one step ahead on a prefetch stream (engine twin); " if pipe else "frozen prefix inside the step; ") +
                                 ("System-1 loss + backward as one hipGraph replay" if not a.no_graph_s1 else "eager") +
                                 ("; frozen prefix as one hipGraph replay per prompt geometry" if not getattr(a, "no_graph_prefix", False) else "") +
                                 ("; the step's own launches on a high-priority stream" if pipe and not getattr(a, "no_priority", False) else ""), "device": arch, "final_loss": round(float(losses[-1].item()), 5)},
            "roofline": {"bound": "mfma", "kernel": dom_name, "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "launches": dom["launches"],
                         "avg_launch_us": round(dom["ms"] * 1e3 / max(dom["launches"], 1), 2), "traffic": None,
                         "kernel_class_ms": {k: round(v["ms"], 3) for k, v in prof.items()},
                         "kernel_class_launches": {k: v["launches"] for k, v in prof.items()},
                         "whole_step": {"algorithmic_tflop": {k: round(v / 1e12, 3) for k, v in fl.items()}, "achieved_tflops_per_gpu": round(step_tflops, 1),
                                        "frac": round(step_tflops / PEAK_BF16_TFLOPS, 4)}},
        }
        line["cpu_baseline"] = cpu_baseline(info, qcfg) if (world == 1 and not a.no_cpu_baseline) else None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
