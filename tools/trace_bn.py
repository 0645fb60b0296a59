"""BatchNorm / elementwise kernels of one replayed step by (name, grid): calls, average and total time (rocprofv3 kernel trace, rocpd .db)
    python tools/trace_bn.py <results.db> [pattern ...]   default: bn_ stem_bn layernorm colsum"""
import sqlite3
import sys
from trace_neighbors import short


def main():
    c = sqlite3.connect(sys.argv[1])
    pats = sys.argv[2:] or ["bn_", "stem_bn", "layernorm", "colsum"]
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    rows = c.execute(f"select name, start, end, {gx or 0} from kernels order by start").fetchall()
    ends = [i for i, r in enumerate(rows) if "sumsq_partial_kernel" in r[0]]
    k = min(range(len(ends) - 1), key=lambda i: rows[ends[i + 1]][2] - rows[ends[i]][2])
    step = rows[ends[k] + 1:ends[k + 1] + 1]
    agg = {}
    for name, s, e, g in step:
        if any(p in name for p in pats):
            a = agg.setdefault((short(name)[:60], g), [0, 0])
            a[0] += 1
            a[1] += e - s
    tot = 0
    for (name, g), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{t / 1e3:8.1f} us  {n:3d} x {t / n / 1e3:6.1f} us  grid {g:8d}  {name}")
        tot += t
    print(f"total {tot / 1e6:.3f} ms")


if __name__ == "__main__":
    main()
