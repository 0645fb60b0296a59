"""Where the waves of each kernel spend their time, from one rocprofv3 PMC pass over an eager step
(--pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM ...):
per kernel name, share of wave cycles parked on s_waitcnt / barriers (WAIT_ANY), stalled at issue (WAIT_INST_ANY), issuing (ACTIVE_INST_ANY).
    python tools/pmc_waves.py results.db [top N]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
by = {}
for k, cn, v, n in rows:
    by.setdefault(k, {})[cn] = v
    by[k]["_n"] = n
steps = max((d["_n"] for k, d in by.items() if "sumsq_partial" in k), default=1)
tot = sum(d.get("SQ_WAVE_CYCLES", 0) for d in by.values())
print(f"{'wave-cycles %':>13s} {'parked':>7s} {'stalled':>8s} {'issuing':>8s} {'VALU/issue':>10s} {'calls/step':>10s}  kernel")
for k, d in sorted(by.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:top]:
    w = d.get("SQ_WAVE_CYCLES", 0) or 1
    name = re.sub(r"_ZN\d*_GLOBAL__N_1|_ZN5clite|clite::|\(anonymous namespace\)::|void ", "", k)[:100]
    print(f"{100 * w / tot:12.1f}% {100 * d.get('SQ_WAIT_ANY', 0) / w:6.0f}% {100 * d.get('SQ_WAIT_INST_ANY', 0) / w:7.0f}% {100 * d.get('SQ_ACTIVE_INST_ANY', 0) / w:7.0f}% "
          f"{100 * d.get('SQ_ACTIVE_INST_VALU', 0) / max(d.get('SQ_ACTIVE_INST_ANY', 1), 1):9.0f}% {d['_n'] / steps:10.1f}  {name}")
