"""One GEMM shape / tile config, a few launches (for rocprofv3 --pmc passes). Usage: python tools/gemm_one.py M N K mode cfg [iters] [group_m]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from internnav_amd import ops  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4])
mode, cfg = sys.argv[4], int(sys.argv[5])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
gm = int(sys.argv[7]) if len(sys.argv) > 7 else 0
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
if mode == "glu":
    out = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
    fn = lambda: ops.linear(x, w, out=out, act="silu", glu=True, force_cfg=cfg, group_m=gm)
elif mode == "f32r":
    out = torch.randn(M, N, device=dev)
    fn = lambda: ops.linear(x, w, out=out, residual=out, force_cfg=cfg, group_m=gm)
else:
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fn = lambda: ops.linear(x, w, out=out, force_cfg=cfg, group_m=gm)
for _ in range(iters):
    fn()
torch.cuda.synchronize()
