"""The non-library nodes of the replayed step from a rocprofv3 trace (kernel + memory-copy): memory copies by (direction, size) and ATen / runtime
kernels by name, per steady-state step. VERDICT r4 next 8.
    rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/cp -o cp -- python3 tools/ab_runtime.py --steps 20
    python tools/copy_nodes.py gpurun_out/cp/cp_results.db 26"""
import collections
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    mt = [t for t in tabs if "memory_cop" in t.lower() or "memcpy" in t.lower()]
    print("tables:", [t for t in tabs if not t.startswith("rocpd_")][:40])
    for t in mt[:2]:
        cols = [r[1] for r in c.execute(f"pragma table_info({t})")]
        print(t, cols)
        size_col = next((x for x in cols if x in ("size", "bytes", "size_bytes")), None)
        kind_col = next((x for x in cols if x in ("name", "kind", "direction", "copy_kind")), None)
        if size_col is None:
            continue
        rows = c.execute(f"select {kind_col or 0}, {size_col}, count(*), sum(end-start) from {t} group by 1, 2 order by 3 desc").fetchall()
        for k, sz, n, tot in rows[:40]:
            print(f"  {n / steps:7.2f} per step  {sz:>12} B  {str(k)[:40]:40s} {tot / max(n, 1) / 1e3:8.1f} us each")
    rows = c.execute("select name, count(*), sum(end-start) from kernels group by name").fetchall()
    other = collections.Counter()
    for n, k, s in rows:
        if "at::native" in n or "rocclr" in n or "elementwise" in n:
            other[n[:150]] += k
    for n, k in other.most_common():
        print(f"  {k / steps:7.2f} per step  kernel {n}")


if __name__ == "__main__":
    main()
