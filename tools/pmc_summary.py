"""Summarise rocprofv3 --pmc output (rocpd database) per kernel: mean of each counter over the dispatches of igemm kernels."""
import collections
import sqlite3
import sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select k.name, p.name, e.value from pmc_events e join pmc_info p on e.pmc_id = p.id join kernels k on e.event_id = k.event_id").fetchall() if False else None
try:
    rows = c.execute("select * from counters_collection limit 1").fetchall()
    cols = [d[0] for d in c.execute("select * from counters_collection limit 1").description]
except Exception as ex:
    print("no counters_collection view:", ex); sys.exit(0)
ci = {n: i for i, n in enumerate(cols)}
rows = c.execute("select * from counters_collection").fetchall()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r[ci["kernel_name"]] if "kernel_name" in ci else r[ci["name"]]
    agg[name[:90]][r[ci["counter_name"]]].append(r[ci["value"]])
for k, d in agg.items():
    if "igemm" not in k and "bn_" not in k:
        continue
    print(k)
    for cn, v in sorted(d.items()):
        print(f"    {cn:32s} n={len(v):3d} mean={sum(v) / len(v):16.1f}")
