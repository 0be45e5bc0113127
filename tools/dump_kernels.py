"""Dump the kernel records of the last `window_ms` of a rocprofv3 (rocpd sqlite) trace as CSV: short name, start / end (us from the window start),
and every other numeric column of the `kernels` view (queue / stream / grid). Usage: python tools/dump_kernels.py <results.db> <out.csv> [window_ms]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
window = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 500e6
tmax = db.execute("select max(end) from kernels").fetchone()[0]
rows = db.execute(f"select * from kernels where start >= {tmax - window} order by start").fetchall()
keep = [i for i, c in enumerate(cols) if c not in (name_col,) and rows and isinstance(rows[0][i], (int, float)) and c not in ("start", "end")]
ni, si, ei = cols.index(name_col), cols.index("start"), cols.index("end")
t0 = rows[0][si] if rows else 0
with open(sys.argv[2], "w") as f:
    f.write("name,start_us,end_us," + ",".join(cols[i] for i in keep) + "\n")
    for r in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", r[ni])
        n = re.sub(r"^void ", "", n).replace(",", ";")[:90]
        f.write(f"{n},{(r[si] - t0) / 1e3:.2f},{(r[ei] - t0) / 1e3:.2f}," + ",".join(str(r[i]) for i in keep) + "\n")
print("columns:", cols)
print("rows:", len(rows))
