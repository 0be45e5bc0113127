"""Run a few graph-replayed steps of the n1_dual bench workload and nothing else (for rocprofv3 --kernel-trace: the trace then ends with timed-step
launches, not with bench.py's eager instrumented pass). Usage: profile_step.py [n_steps] [bench flags, e.g. --no-row-chain]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3
kw = {}
for i, x in enumerate(sys.argv):
    if x == "--dit-ffn":
        kw["dit_ffn"] = int(sys.argv[i + 1])
    elif x in ("--no-row-chain", "--no-fuse-decode-rope", "--no-frag-weights", "--no-split-prefill"):
        kw[x[2:].replace("-", "_")] = True
wl = bench.N1Dual(bench.default_args(**kw), torch.device("cuda:0"), 0)
wl.capture()
for i in range(2 + n):
    wl.step(i)
torch.cuda.synchronize()
print("done", n, kw)
