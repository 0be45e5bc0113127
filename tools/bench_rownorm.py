"""Row-block GEMM + gated-norm epilogue vs the GEMM + chained-norm pair on the DiT shapes. Usage: python tools/bench_rownorm.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from internnav_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, N, div = 65536, 384, 1024


def timeit(fn, iters=30):
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


for K in (384, 1024):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    x = torch.randn(M, N, device=dev)
    h = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    pj = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    g1, g2 = torch.ones(N, device=dev), torch.ones(N, device=dev)
    mod = torch.randn(M // div, 2 * N, device=dev) * 0.1
    gate, ms = mod[:, :N], mod[:, N:]
    fused = lambda: ops.gemm_rownorm(a, w, g1, x, gate=gate, h=h, gamma2=g2, mod_scale2=ms, mod_div=div)

    def pair():
        ops.linear(a, w, out=pj)
        ops.norm(pj, g1, None, eps=1e-5, rms=True, gate=gate, base=x, mod_div=div, out32=x, out2=h, gamma2=g2, mod_scale2=ms)

    tf, tp = timeit(fused), timeit(pair)
    print(f"K={K:5d}: fused {tf:7.1f} us ({2.0 * M * N * K / tf / 1e6:6.1f} TF/s)   GEMM + chained norm {tp:7.1f} us")
