"""Where does the decode chain lose its time beside System-1? From a rocprofv3 kernel trace (rocpd sqlite) of a few bench steps: for the
weight-streaming decode kernels of the LAST recorded step - their durations and the idle gaps between consecutive kernels of the chain (same
stream) - next to the same quantities for the chain running alone (a second db, e.g. of `tools/profile_phases.py s2`), if given.
Usage: python tools/phase_timeline.py <bench_results.db> [alone_results.db]"""
import re
import sqlite3
import sys
from collections import defaultdict


def load(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    qcol = next((c for c in cols if c in ("queue_id", "stream_id", "queue")), None)
    gcol = [c for c in cols if c.startswith("grid") or c.startswith("workgroup")]
    sel = f"select {name_col}, start, end" + (f", {qcol}" if qcol else ", 0") + "".join(f", {c}" for c in gcol) + " from kernels order by start"
    rows = db.execute(sel).fetchall()
    return rows, cols, gcol


def chain_stats(rows, label):
    # the decode chain = the queue that runs the skinny GEMMs; take the last contiguous burst of >= 500 of them
    sk = [i for i, r in enumerate(rows) if "gemm_skinny" in r[0]]
    if not sk:
        print(label, "no skinny GEMMs in trace")
        return
    q = rows[sk[-1]][3]
    # walk back from the last skinny kernel over kernels of the same queue until a gap > 20 ms (previous step)
    last = sk[-1]
    chain = []
    i = last
    prev_start = rows[last][1]
    while i >= 0:
        r = rows[i]
        if r[3] == q:
            if prev_start - r[2] > 20e6:
                break
            chain.append(r)
            prev_start = r[1]
        i -= 1
    chain.reverse()
    # restrict to the part from the first skinny kernel with M<=16-style chain (after the prefill's tiled GEMMs): start at first skinny
    first = next(k for k, r in enumerate(chain) if "gemm_skinny" in r[0])
    chain = chain[first:]
    t0, t1 = chain[0][1], chain[-1][2]
    busy = sum(r[2] - r[1] for r in chain)
    gaps = [max(0, chain[k + 1][1] - chain[k][2]) for k in range(len(chain) - 1)]
    print(f"{label}: {len(chain)} kernels on the decode queue, span {(t1 - t0) / 1e6:.2f} ms, kernel time {busy / 1e6:.2f} ms, idle gaps {sum(gaps) / 1e6:.2f} ms "
          f"(median gap {sorted(gaps)[len(gaps) // 2] / 1e3:.1f} us, p90 {sorted(gaps)[int(len(gaps) * 0.9)] / 1e3:.1f} us)")
    by = defaultdict(list)
    for r in chain:
        n = re.sub(r"\(anonymous namespace\)::", "", r[0])
        n = re.sub(r"^void ", "", n)[:70]
        by[(n,) + tuple(r[4:])].append((r[2] - r[1]) / 1e3)
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:14]:
        v.sort()
        print(f"   {k[0]:70s} grid {k[1:]}  x{len(v):5d}  avg {sum(v) / len(v):8.1f} us  median {v[len(v) // 2]:8.1f}  p90 {v[int(len(v) * 0.9)]:8.1f}  total {sum(v) / 1e3:7.2f} ms")


def main():
    rows, cols, gcol = load(sys.argv[1])
    print("# kernels table columns:", cols)
    chain_stats(rows, "in the step (beside System-1)")
    if len(sys.argv) > 2:
        rows2, _, _ = load(sys.argv[2])
        chain_stats(rows2, "alone")


if __name__ == "__main__":
    main()
