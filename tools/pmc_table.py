"""Generic per-kernel table of a rocprofv3 --pmc pass: calls, average duration and every collected counter (sum over launches / launches).
Usage: python tools/pmc_table.py <results.db> [top_n] [name_filter]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
flt = sys.argv[3] if len(sys.argv) > 3 else ""
rows = db.execute("select kernel_name, counter_name, count(*), sum(value), sum(duration) from counters_collection group by kernel_name, counter_name").fetchall()
K, names = {}, []
for k, c, n, v, d in rows:
    e = K.setdefault(k, {"n": n, "dur": d})
    e[c] = v
    if c not in names:
        names.append(c)


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n if len(n) <= 60 else n[:57] + "..."


keys = [k for k in sorted(K, key=lambda k: -(K[k]["dur"] or 0)) if flt in k][:top]
print(f"{'kernel':60s} {'calls':>6s} {'avg_us':>8s} " + " ".join(f"{c[:22]:>22s}" for c in names))
for k in keys:
    e = K[k]
    print(f"{short(k):60s} {e['n']:6d} {e['dur'] / e['n'] / 1e3:8.1f} " + " ".join(f"{(e.get(c, float('nan')) / e['n']):22.4g}" for c in names))
if "SQ_WAVE_CYCLES" in names:
    print("\n# fractions of wave time (quad-cycle counters): parked = SQ_WAIT_ANY, issue-stalled = SQ_WAIT_INST_ANY, issuing = SQ_ACTIVE_INST_ANY")
    for k in keys:
        e = K[k]
        wc = e.get("SQ_WAVE_CYCLES") or float("nan")
        gui = e.get("GRBM_GUI_ACTIVE")
        mf = e.get("SQ_VALU_MFMA_BUSY_CYCLES")
        busy = (mf / 1024.0) / (gui / 8.0) if gui and mf is not None else float("nan")
        clk = (gui / 8.0) / (e["dur"] * 1e-9) / 1e9 if gui else float("nan")
        print(f"{short(k):60s} parked {e.get('SQ_WAIT_ANY', float('nan')) / wc:6.3f} stalled {e.get('SQ_WAIT_INST_ANY', float('nan')) / wc:6.3f} "
              f"issuing {e.get('SQ_ACTIVE_INST_ANY', float('nan')) / wc:6.3f} mfma_busy {busy:6.3f} eff_clock_GHz {clk:5.2f} "
              f"waves_resident_per_simd {wc * 4 / (gui / 8.0) / 1024 if gui else float('nan'):5.2f}")
