"""Dump every counter of a rocprofv3 --pmc run per kernel (sum over dispatches / number of dispatches). Usage: pmc_dump.py <db> [name filter]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
for k, c, n, v in rows:
    if flt in k:
        print(f"{k[:70]:70s} {c:32s} n={n:4d} avg={v / n:16.1f}")
