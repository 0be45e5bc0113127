"""Chained pre-norm vs two launches on the DiT shape (65536 x 384). Usage: python tools/bench_norm_chain.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from internnav_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
rows, C, div = 65536, 384, 1024


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


proj = torch.randn(rows, C, device=dev).to(torch.bfloat16)
x = torch.randn(rows, C, device=dev)
h = torch.empty(rows, C, device=dev, dtype=torch.bfloat16)
g1, g2 = torch.ones(C, device=dev), torch.ones(C, device=dev)
mod = torch.randn(rows // div, 4 * C, device=dev) * 0.1
gate, ms = mod[:, :C], mod[:, C:2 * C]
a = lambda: ops.norm(proj, g1, None, eps=1e-5, rms=True, gate=gate, base=x, mod_div=div, out32=x)
b = lambda: ops.norm(x, g2, None, eps=1e-5, rms=True, mod_scale=ms, mod_div=div, out=h)
c = lambda: ops.norm(proj, g1, None, eps=1e-5, rms=True, gate=gate, base=x, mod_div=div, out32=x, out2=h, gamma2=g2, mod_scale2=ms)
ta, tb, tc = timeit(a), timeit(b), timeit(c)
mb = rows * C / 1e6
print(f"gated residual norm     {ta:7.1f} us  {mb * 10 / ta * 1e-0:8.1f} MB/us-> {mb * 10 / ta / 1e3:.2f} TB/s")
print(f"modulated pre-norm      {tb:7.1f} us  {mb * 6 / tb / 1e3:.2f} TB/s")
print(f"chained (one launch)    {tc:7.1f} us  {mb * 12 / tc / 1e3:.2f} TB/s   vs separate {ta + tb:7.1f} us")
