"""Micro-benchmarks of the op-level kernels on the GPU box: TFLOP/s per GEMM shape / tile config, attention, norm GB/s.
Usage: python tools/bench_ops.py [--out gpurun_out/bench_ops.json]
"""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from internnav_amd import ops  # noqa: E402


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/bench_ops.json")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    res = {"gemm": [], "attn": [], "norm": []}
    shapes = [
        (49152, 1152, 384), (49152, 384, 384), (49152, 1536, 384), (49152, 384, 1536),   # NavDP decoder @B=64
        (148032, 1152, 384), (148032, 1536, 384), (148032, 384, 1536),                  # DINOv2 576 images
        (65536, 384, 1024), (65536, 2048, 384),
        (3136 * 16, 3840, 1280), (3136 * 16, 1280, 1280), (3136 * 16, 6848, 1280), (3136 * 16, 1280, 3424),  # Qwen ViT, 16 envs
        (920 * 16, 4608, 3584), (920 * 16, 3584, 3584), (920 * 16, 37888, 3584), (920 * 16, 3584, 18944),   # LLM prefill, 16 envs
        (64, 4608, 3584), (64, 37888, 3584), (64, 3584, 18944), (64, 152064, 3584),                         # decode, 64 envs
        (4096, 4096, 4096), (8192, 8192, 8192),
    ]
    for (M, N, K) in shapes:
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        row = {"M": M, "N": N, "K": K}
        for cfg in ([0, 1, 2, 5] if M > 64 else [0, 3, 4]):
            try:
                t = timeit(lambda: ops.linear(x, w, out=out, force_cfg=cfg), iters=10)
                row[f"cfg{cfg}_tflops"] = round(2.0 * M * N * K / t * 1e-12, 1)
            except Exception as ex:  # noqa
                row[f"cfg{cfg}_err"] = str(ex)[:100]
        res["gemm"].append(row)
        print(row, flush=True)
        del x, w, out
    cases = [
        ("dinov2", 576, 257, 257, 6, 6, 64, False, 1), ("navdp_self", 2048, 24, 24, 8, 8, 48, True, 1),
        ("navdp_cross", 2048, 24, 132, 8, 8, 48, False, 32), ("former", 64, 128, 2304, 8, 8, 48, False, 1),
        ("dit_self", 2048, 32, 32, 6, 6, 64, False, 1), ("qwen_vit_full", 64, 784, 784, 16, 16, 80, False, 1),
        ("llm_prefill", 16, 920, 920, 28, 4, 128, True, 1),
    ]
    for (name, B, Lq, Lk, H, Hkv, D, causal, bdiv) in cases:
        q = torch.randn(B, Lq, H, D, device=dev).to(torch.bfloat16)
        k = torch.randn(B // bdiv, Lk, Hkv, D, device=dev).to(torch.bfloat16)
        v = torch.randn(B // bdiv, Lk, Hkv, D, device=dev).to(torch.bfloat16)
        o = torch.empty_like(q)
        t = timeit(lambda: ops.attention(q, k, v, causal=causal, kv_bdiv=bdiv, out=o), iters=10)
        fl = 4.0 * B * H * Lq * Lk * D * (0.5 if causal else 1.0)
        row = {"name": name, "ms": round(t * 1e3, 4), "tflops": round(fl / t * 1e-12, 2)}
        res["attn"].append(row)
        print(row, flush=True)
    for rows, C in [(49152, 384), (148032, 384), (14720, 3584), (50176, 1280)]:
        x = torch.randn(rows, C, device=dev).to(torch.bfloat16)
        gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        y = torch.empty_like(x)
        t = timeit(lambda: ops.norm(x, gamma=gam, beta=bet, out=y), iters=20)
        row = {"rows": rows, "C": C, "us": round(t * 1e6, 1), "GBps": round(4.0 * rows * C / t * 1e-9, 1)}
        res["norm"].append(row)
        print(row, flush=True)
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    Path(a.out).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
