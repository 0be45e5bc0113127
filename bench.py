#!/usr/bin/env python
"""Benchmark of the MI355X policy engine: policy steps / sec / node (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--envs 64] [--workload navdp_s1]
  N > 1 is launched by the driver as  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Workload at N = 1 (config.workload):  "navdp_s1_b64" = BASELINE config #2 - NavDP System-1 only: NavDPNet diffusion
trajectory head, 10 DDPM steps x 32 samples + critic ranking, batch = 64 envs per GPU, 8 RGB frames + 1 depth frame of
224x224 per env, seeded random weights at the true architecture shapes, synthetic frames. One bench "step" = one
policy step of every env of the rank (one `predict_pointgoal_batch_action_vel` over 64 envs) = 64 policy steps.
Episodes are independent, so ranks shard envs with no data-path exchange ("scaling": "weak", per-GPU work fixed); the
only collective is the all_gather of the per-env action outputs over RCCL/xGMI (north_star) once per step.

Timed region: inputs resident in HBM, W warm-up steps, then exactly K steps bracketed by barrier + device synchronise,
max over ranks. The step replays one hipGraph of the whole call (+ device-side noise draw + the all_gather when N > 1).
After the timed region rank 0 runs ONE instrumented eager pass with per-launch HIP events (ina_prof_*) to attribute
time and algorithmic FLOPs to kernel classes for the "roofline" object, and (N = 1 only) times the CPU oracle on a
bounded sample for "cpu_baseline".
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16 peak of MI355X (guides/MI355X_MICROARCH.md: ~2.5 PF dense)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs", type=int, default=64, help="environments per GPU")
    ap.add_argument("--workload", default="navdp_s1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay (debug)")
    return ap.parse_args()


def cpu_baseline(cfg, seed=0):
    """The reference PyTorch path on the host cores: the CPU oracle (a port of the reference's NavDPNet inference, pinned
    against the reference modules) timed on a BOUNDED sample: one env, one full policy step, fp32, batch-1 as the reference runs."""
    from internnav_amd import synthetic
    from oracle import navdp as o_navdp  # cpu_baseline leg only

    cores = min(64, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    sd = synthetic.navdpnet_state_dict(seed)
    inp = synthetic.navdpnet_inputs(1, seed)
    with torch.no_grad():
        t0 = time.time()
        o_navdp.navdpnet_pointgoal(sd, inp["goal"], inp["images"], inp["depths"], inp["x_init"], inp["step_noise"], cfg)
        dt = time.time() - t0
    return {"value": round(1.0 / dt, 4), "unit": "policy steps/s", "cores": cores, "kind": "port",
            "sample": "1 env x 1 policy step (9 ViT-S frames + 10 DDPM steps x 32 samples + critic), fp32 torch CPU, batch-1 as the reference executes",
            "seconds": round(dt, 2)}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run --nproc-per-node N"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from internnav_amd import flops, runtime, synthetic
    from internnav_amd.navdp import NavDPNet

    arch = runtime.require_gfx950()
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
    assert a.workload == "navdp_s1", "only the NavDP System-1 workload (BASELINE config #2) is benchmarked in this round"
    cfg = synthetic.NAVDPNET_CFG
    B = a.envs
    sd = synthetic.navdpnet_state_dict(seed=0)
    net = NavDPNet(sd, cfg, dev, max_envs=B)
    del sd
    # synthetic inputs generated on the device (seed = 1000*rank): frames in 0..1, depth in metres, point goals
    g = torch.Generator(device=dev).manual_seed(1000 * rank + 7)
    inp = dict(
        goal=torch.randn(B, 3, device=dev, generator=g) * 3.0,
        images=torch.rand(B, cfg["memory_size"], 224, 224, 3, device=dev, generator=g),
        depths=torch.rand(B, 1, 224, 224, 1, device=dev, generator=g) * 5.0,
        x_init=torch.randn(B, cfg["sample_num"], cfg["predict_size"], 3, device=dev, generator=g),
        step_noise=torch.randn(cfg["num_train_timesteps"], B, cfg["sample_num"], cfg["predict_size"], 3, device=dev, generator=g),
    )

    def call(goal, images, depths, x_init, step_noise):
        return net.predict_pointgoal_batch_action_vel(goal, images, depths, x_init, step_noise)

    if a.no_graph:
        def run():
            return call(**inp)
    else:
        graphed = runtime.GraphedCall(call, inp)

        def run():
            return graphed()

    gathered = None
    if world > 1:
        gathered = torch.empty(world * B, 8, cfg["predict_size"], 3, device=dev)

    def step():
        # fresh sampler noise every policy step (drawn on the device into the static buffers the graph reads)
        inp["x_init"].normal_(generator=g)
        inp["step_noise"].normal_(generator=g)
        neg, pos = run()
        if world > 1:
            dist.all_gather_into_tensor(gathered, pos)  # per-env action outputs to every rank (RCCL over xGMI)
        return pos

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    value = world * B * a.steps / dt

    if rank == 0:
        # ---- roofline: one instrumented eager pass, HIP events around every launch on the launch stream
        f_alg = flops.navdpnet_flops_per_env(cfg)
        runtime.prof_enable(True)
        call(**inp)
        torch.cuda.synchronize()
        prof = runtime.prof_read()
        runtime.prof_enable(False)
        gm = prof["gemm"]
        kernel_ms = {k: round(v["ms"], 3) for k, v in prof.items()}
        achieved = gm["flops"] / (gm["ms"] * 1e-3) / 1e12 if gm["ms"] > 0 else 0.0
        step_tflops = (value / world) * f_alg["total"] / 1e12
        roofline = {
            "bound": "mfma", "kernel": "gemm_bf16_nt_kernel (all tile configs)",
            "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
            "traffic": None,
            "launches_per_step": gm["launches"], "avg_launch_us": round(gm["ms"] * 1e3 / max(gm["launches"], 1), 2),
            "gemm_flops_per_step": gm["flops"], "kernel_class_ms_per_step": kernel_ms,
            "whole_step": {"algorithmic_tflop_per_env_step": round(f_alg["total"] / 1e12, 4),
                           "achieved_tflops_per_gpu": round(step_tflops, 1), "frac": round(step_tflops / PEAK_BF16_TFLOPS, 4)},
        }
        line = {
            "metric": "policy steps/sec/node", "value": round(value, 2), "unit": "policy steps/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"navdp_s1_b{B}", "policy": "NavDPNet (BASELINE config #2: NavDP System-1 only)",
                       "envs_per_gpu": B, "samples_per_env": cfg["sample_num"], "ddpm_steps": cfg["num_train_timesteps"],
                       "frames_per_env": f"{cfg['memory_size']} rgb + 1 depth @224x224", "parallelism": f"dp{world}",
                       "launch": "eager" if a.no_graph else "hipGraph replay", "device": arch},
            "roofline": roofline,
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
