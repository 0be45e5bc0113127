"""Weight-streaming rate of the skinny (M <= 64) GEMM path on the decode shapes. Usage: python tools/bench_skinny.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from internnav_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


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


for (M, N, K, glu) in [(7, 4608, 3584, 0), (7, 3584, 3584, 0), (7, 37888, 3584, 1), (7, 3584, 18944, 0), (7, 152064, 3584, 0), (35, 37888, 3584, 1), (64, 18816, 384, 0)]:
    # rotate over several weight copies so the 256 MiB infinity cache cannot serve the stream
    nrep = max(1, int(600e6 // (N * K * 2)) + 1)
    ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(min(nrep, 8))]
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N // 2 if glu else N, device=dev, dtype=torch.bfloat16)
    i = [0]

    row = f"M={M:3d} N={N:6d} K={K:6d} glu={glu}:"
    for cfg, name in ((31, "split-K + epilogue kernel"), (32, "column-owner fused")):
        def fn():
            i[0] = (i[0] + 1) % len(ws)
            ops.linear(x, ws[i[0]], out=out, act="silu" if glu else None, glu=bool(glu), force_cfg=cfg)

        t = timeit(fn)
        row += f"   {name}: {t*1e6:7.1f} us {N*K*2/t*1e-12:5.2f} TB/s"
    print(row)
lg = torch.randn(7, 152064, device=dev)
o = torch.empty(7, dtype=torch.int32, device=dev)
print(f"argmax 7 x 152064: {timeit(lambda: ops.argmax_rows(lg, o))*1e6:.1f} us")
