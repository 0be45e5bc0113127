"""GEMM tile-config sweep on the hot shapes (MI355X). Usage: python tools/bench_gemm.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from internnav_amd import ops  # noqa: E402


def timeit(fn, iters=int(__import__("os").environ.get("GEMM_ITERS", "10")), warmup=3):
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


dev = torch.device("cuda:0")
import os
ACT = int(os.environ.get('GEMM_ACT', '0'))
shapes = [
    (5520, 4608, 3584, "bf16"), (5520, 3584, 3584, "f32r"), (5520, 37888, 3584, "glu"), (5520, 3584, 18944, "f32r"),
    (21952, 6912, 1280, "glu"), (21952, 1280, 3456, "f32r"), (18816, 3840, 1280, "bf16"), (18816, 6912, 1280, "glu"),
    (6440, 4608, 3584, "bf16"), (6440, 3584, 3584, "f32r"), (6440, 37888, 3584, "glu"), (6440, 3584, 18944, "f32r"),
    (21952, 3840, 1280, "bf16"), (21952, 1280, 1280, "f32r"), (21952, 6848, 1280, "glu"), (21952, 1280, 3424, "f32r"),
    (65536, 1536, 384, "bf16"), (65536, 384, 384, "f32r"), (65536, 2048, 384, "glu"), (65536, 384, 1024, "f32r"),
    (49152, 1152, 384, "bf16"), (49152, 384, 1536, "f32r"), (8192, 8192, 8192, "bf16"),
]
cfgs = [int(c) for c in sys.argv[1:] if not c.startswith('s')] or [1, 6, 7, 8]
if 's1' in sys.argv:
    shapes = [s for s in shapes if s[0] >= 49152]
for (M, N, K, mode) in shapes:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    row = f"{M:6d} {N:6d} {K:6d} {mode:5s}"
    for cfg in cfgs:
        if mode == "bf16":
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            fn = lambda: ops.linear(x, w, out=out, force_cfg=cfg, act=ACT)
        elif mode == "glu":
            out = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
            fn = lambda: ops.linear(x, w, out=out, act="silu", glu=True, force_cfg=cfg)
        else:
            out = torch.randn(M, N, device=dev)
            fn = lambda: ops.linear(x, w, out=out, residual=out, force_cfg=cfg)
        try:
            t = timeit(fn)
            row += f"  cfg{cfg}: {2.0 * M * N * K / t * 1e-12:7.1f} TF"
        except Exception as ex:  # noqa
            row += f"  cfg{cfg}: ERR {str(ex)[:40]}"
    print(row, flush=True)
