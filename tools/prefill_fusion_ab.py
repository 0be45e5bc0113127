"""Timing-only A/B asked for by the round-5 review (item 6): the 7-env System-2 prefill graph with its RMSNorm and / or rope + KV-append launches
stubbed out (WRONG results on purpose) - the upper bound of what fusing them into the neighbouring GEMMs could buy inside the two-stream prefill.
Usage (GPU box): python tools/prefill_fusion_ab.py"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from internnav_amd import ops as real_ops  # noqa: E402
from internnav_amd import qwen_vl, runtime  # noqa: E402


class Stub:
    """`ops` with selected launches dropped for calls over many rows (the prefill; single-token passes are not touched)"""

    def __init__(self, drop):
        self.drop = set(drop)

    def __getattr__(self, name):
        f = getattr(real_ops, name)
        if name not in self.drop:
            return f

        def g(x, *a, **kw):
            rows = kw.get("rows") or x.shape[0]
            if rows > 64:
                return kw.get("out", x)
            return f(x, *a, **kw)
        return g


def t(fn, n=5):
    fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


dev = torch.device("cuda:0")
wl = bench.N1Dual(bench.default_args(), dev, 0)
q = wl.model.qwen
m = max(wl.mb)
s = wl.s2[m]
wl._ingest_s2(0, m, s["pv"])
for name, drop in (("all launches", ()), ("no RMSNorm launches", ("norm",)), ("no rope + KV-append launches", ("rope",)), ("neither", ("norm", "rope")), ("all launches (again)", ())):
    qwen_vl.ops = Stub(drop)
    g = runtime.GraphedCall(lambda: q.run_prefill(s["P"], s["pv"]), {})
    print(f"prefill graph, {m} envs, {name:32s} {t(g):7.2f} ms", flush=True)
qwen_vl.ops = real_ops
