"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; values are KiB per dispatch).
gfx950 correction (guides/MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-reports wide coalesced streaming reads by exactly 2x
(128-byte requests tallied at 64 B) -> reads are doubled; WRITE_SIZE is uncalibrated and reported as is.
Usage: python tools/pmc_summary.py <fetch.db> <write.db> [top_n]"""
import re
import sqlite3
import sys


def load(path, name):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, count(*), sum(value), sum(duration) from counters_collection where counter_name = ? "
                      "group by kernel_name", (name,)).fetchall()
    return {r[0]: r[1:] for r in rows}


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n if len(n) <= 90 else n[:87] + "..."


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
keys = sorted(f, key=lambda k: -(f[k][2] or 0))[:top]
print(f"{'kernel':90s} {'calls':>7s} {'read MB/launch (x2 corrected)':>30s} {'write MB/launch':>16s} {'avg_us':>8s} {'HBM TB/s':>9s}")
for k in keys:
    c, kib, dur = f[k]
    wk = w.get(k, (c, 0, dur))
    rd = 2.0 * kib * 1024 / c / 1e6
    wr = wk[1] * 1024 / max(wk[0], 1) / 1e6
    us = dur / c / 1e3
    print(f"{short(k):90s} {c:7d} {rd:30.2f} {wr:16.2f} {us:8.1f} {(rd + wr) / us:9.2f}")
