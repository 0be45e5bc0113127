"""dit_ffn (fused SwiGLU FFN + gated norm + residual + next pre-norm) vs the three launches it replaces, M = 64 envs x 32 samples x 32 tokens.
Usage: python tools/bench_ffn.py [M]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from internnav_amd import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
D, F, div = 384, 1024, 1024
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
h = torch.randn(M, D, device=dev, generator=g).to(torch.bfloat16)
w13 = (torch.randn(2 * F, D, device=dev, generator=g) * D ** -0.5).to(torch.bfloat16)
w2 = (torch.randn(D, F, device=dev, generator=g) * F ** -0.5).to(torch.bfloat16)
x = torch.randn(M, D, device=dev, generator=g)
g1 = torch.ones(D, device=dev)
mod = 0.1 * torch.randn(M // div, 2 * D, device=dev, generator=g)
ff = torch.empty(M, F, dtype=torch.bfloat16, device=dev)
pj = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
h2 = torch.empty(M, D, dtype=torch.bfloat16, device=dev)


def fused():
    ops.dit_ffn(h, w13, w2, g1, x, gate=mod[:, :D], h=h2, gamma2=g1, mod_scale2=mod[:, D:], mod_div=div)


def variant(flags):
    return lambda: ops.dit_ffn(h, w13, w2, g1, x, gate=mod[:, :D], h=h2, gamma2=g1, mod_scale2=mod[:, D:], mod_div=div, rotate=flags)


def unfused():
    ops.linear(h, w13, act="silu", glu=True, out=ff)
    ops.linear(ff, w2, out=pj)
    ops.norm(pj, g1, None, eps=1e-5, rms=True, gate=mod[:, :D], base=x, mod_div=div, out32=x, out2=h2, gamma2=g1, mod_scale2=mod[:, D:])


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


fl = 2.0 * M * D * F * 3
for name, fn in (("fused dit_ffn", fused), ("  dbg: rotated chunk order", variant(1)), ("  dbg: weights loaded once", variant(2)), ("  dbg: no x / h epilogue I/O", variant(4)),
                 ("  dbg: neither", variant(6)), ("glu gemm + gemm + norm", unfused)):
    us = timeit(fn)
    print(f"{name:28s} M={M}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s")
