"""Print PMC counters per kernel from a rocprofv3 rocpd database: python tools/pmc_dump.py results.db [name-substring]"""
import sqlite3
import sys
c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
by = {}
for k, cn, v, n in rows:
    if pat in k:
        by.setdefault(k[:110], {})[cn] = (v, n)
for k, d in by.items():
    print(k)
    for cn, (v, n) in sorted(d.items()):
        print(f"    {cn:28s} {v / n:16.0f} per dispatch ({n} dispatches)")
