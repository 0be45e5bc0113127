"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace: per-kernel calls / total / average duration.
Usage: python tools/rocprof_summary.py <results.db> [top_n]   (writes a text table to stdout; committed under profiles/)."""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= 110 else name[:107] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {name_col}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                      f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"{'kernel':110s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>9s} {'%':>6s}")
    for n, c, s, a, mn, mx in rows[:top]:
        print(f"{short(n):110s} {c:8d} {s / 1e6:10.3f} {a / 1e3:9.2f} {mn / 1e3:8.2f} {mx / 1e3:9.2f} {100.0 * s / total:6.2f}")
    print(f"{'TOTAL':110s} {sum(r[1] for r in rows):8d} {total / 1e6:10.3f}")


if __name__ == "__main__":
    main()
