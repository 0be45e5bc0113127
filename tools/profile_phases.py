"""Run N eager S2 micro-batch calls or N eager S1 calls of the n1_dual workload (for rocprofv3 --kernel-trace). Usage: profile_phases.py s2|s1 [n]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

which, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3
wl = bench.N1Dual(bench.default_args(no_overlap=True, no_graph=True), torch.device("cuda:0"), 0)
m = max(wl.mb)
wl.model.qwen.split_serial = True     # the timed step's launches (two half micro-batches, shared-tail tile selection) on ONE stream: per-kernel durations of kernels that do not overlap (as bench.py's instrumented pass)
wl._ingest_s2(0, m, wl.s2[m]["pv"])
for _ in range(n):
    if which == "s2":
        wl._s2_call(m)
    else:
        wl._s1_call()
torch.cuda.synchronize()
print("done", which, n)
