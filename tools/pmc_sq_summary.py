"""Per-kernel MFMA utilisation from a rocprofv3 --pmc SQ pass (counters_collection table of the results .db).
Usage: python tools/pmc_sq_summary.py <results.db> [top_n]
Units (guides/MI355X_MICROARCH.md, PMC section; checked in round 1 on the 8192^3 GEMM): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles
summed over the 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE counts cycles summed over the 8 XCDs; SQ_WAVE_CYCLES / SQ_WAIT_INST_LDS / SQ_BUSY_CYCLES
count quad-cycles. So
  mfma_busy      = (SQ_VALU_MFMA_BUSY_CYCLES / 1024) / (GRBM_GUI_ACTIVE / 8)      fraction of the kernel's cycles a SIMD's matrix core is busy
  lds_wait       = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES                               fraction of wave time spent waiting on LDS instructions
  implied TF/s   = mfma_busy x 2500 (dense bf16 peak)   -> compare with algorithmic FLOPs / duration of the same kernel
"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = db.execute("select kernel_name, counter_name, count(*), sum(value), sum(duration) from counters_collection group by kernel_name, counter_name").fetchall()
K = {}
for k, c, n, v, d in rows:
    e = K.setdefault(k, {"n": n, "dur": d})
    e[c] = v


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n if len(n) <= 72 else n[:69] + "..."


keys = sorted(K, key=lambda k: -(K[k]["dur"] or 0))[:top]
print(f"{'kernel':72s} {'calls':>6s} {'avg_us':>8s} {'total_ms':>9s} {'mfma_busy':>10s} {'implied_TF/s':>12s} {'lds_wait':>9s} {'sq_busy':>8s}")
for k in keys:
    e = K[k]
    gui = e.get("GRBM_GUI_ACTIVE")
    mf = e.get("SQ_VALU_MFMA_BUSY_CYCLES")
    busy = (mf / 1024.0) / (gui / 8.0) if gui and mf is not None else float("nan")
    wc = e.get("SQ_WAVE_CYCLES")
    lw = e.get("SQ_WAIT_INST_LDS")
    ldsw = lw / wc if wc and lw is not None else float("nan")
    sb = e.get("SQ_BUSY_CYCLES")
    sqb = (sb * 4 / 8.0) / (gui / 8.0) / 8.0 if gui and sb is not None else float("nan")   # quad-cycles, per-XCD SE aggregate: indicative only
    print(f"{short(k):72s} {e['n']:6d} {e['dur'] / e['n'] / 1e3:8.1f} {e['dur'] / 1e6:9.2f} {busy:10.3f} {busy * 2500:12.0f} {ldsw:9.3f} {sqb:8.2f}")
