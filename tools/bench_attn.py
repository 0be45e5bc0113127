"""Attention forward on the hot shapes (MI355X). Usage: python tools/bench_attn.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from internnav_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def dense(name, B, L, H, Hkv, D, causal):
    q = torch.randn(B, L, H, D, device=dev).to(torch.bfloat16)
    k = torch.randn(B, L, Hkv, D, device=dev).to(torch.bfloat16)
    v = torch.randn(B, L, Hkv, D, device=dev).to(torch.bfloat16)
    o = torch.empty_like(q)
    us = timeit(lambda: ops.attention(q, k, v, causal=causal, out=o, kernel=KERNEL))
    fl = 4.0 * B * H * L * L * D * (0.5 if causal else 1.0)
    print(f"{name:34s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s")


KERNEL = 0

for KERNEL in (1, 2):
    print("== ina_attn_args.kernel =", KERNEL, "(1 = 16-rows-per-wave kernel, 2 = 32-rows-per-wave kernel)")
    dense("LLM prefill 7x920 28/4 x128 causal", 7, 920, 28, 4, 128, True)
    dense("LLM prefill 6x920 28/4 x128 causal", 6, 920, 28, 4, 128, True)
    dense("7x920 28/4 x128 NOT causal", 7, 920, 28, 4, 128, False)
    dense("ViT full 28x784 16 x80", 28, 784, 16, 16, 80, False)
    dense("DINOv2 128x257 6 x64", 128, 257, 6, 6, 64, False)
KERNEL = 0
# ViT windows: 64-token windows packed on one axis
Np, H, D = 21952, 16, 80
qkv = torch.randn(Np, 3, H, D, device=dev).to(torch.bfloat16)
cu = torch.arange(0, Np + 1, 64, dtype=torch.int32, device=dev)
o = torch.empty(Np, H, D, device=dev, dtype=torch.bfloat16)
us = timeit(lambda: ops.attention(qkv[:, 0], qkv[:, 1], qkv[:, 2], cu_q=cu, cu_k=cu, max_q=64, max_k=64, out=o))
print(f"{'ViT windows 343x64 16 x80':34s} {us:8.1f} us  {4.0 * (Np // 64) * H * 64 * 64 * D / us / 1e6:7.1f} TF/s")
# decode: 1 and 5 query tokens against the KV cache (GQA-packed split-KV kernel + merge)
for Lq in (1, 5):
    B, Lk, H, Hkv, D = 7, 927 + Lq, 28, 4, 128
    q = torch.randn(B, Lq, H, D, device=dev).to(torch.bfloat16)
    k = torch.randn(B, Lk, Hkv, D, device=dev).to(torch.bfloat16)
    v = torch.randn(B, Lk, Hkv, D, device=dev).to(torch.bfloat16)
    o = torch.empty_like(q)
    us = timeit(lambda: ops.attention(q, k, v, causal=True, out=o), iters=50)
    print(f"{'decode 7 x ' + str(Lq) + ' vs ' + str(Lk) + ' keys 28/4 x128':34s} {us:8.1f} us  ({2.0 * B * Hkv * Lk * D * 2 * 2 / us / 1e6:5.2f} TB/s of KV)")
