"""Per-kernel HBM-side traffic from the two PMC passes of tools/pmc_traffic.py (same databases): GB fetched (x2 corrected) and written per
step for every kernel name, largest first.   python tools/pmc_by_kernel.py <fetch .db> <write .db> [top]"""
import collections
import re
import sqlite3
import sys


def table(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name = ? group by kernel_name", (counter,)).fetchall()
    steps = max((n for name, _, n in rows if "sgd_step_kernel" in name), default=1)
    return {name: (val * 1024.0 / steps / 1e9, n / steps) for name, val, n in rows}, steps


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|clite::|_GLOBAL__N_1", "", name)
    return name[:150]


def main():
    f, sf = table(sys.argv[1], "FETCH_SIZE")
    w, sw = table(sys.argv[2], "WRITE_SIZE")
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    names = sorted(set(f) | set(w), key=lambda n: -(2 * f.get(n, (0, 0))[0] + w.get(n, (0, 0))[0]))
    print(f"steps in pass: {sf} / {sw}")
    print(f"{'fetch GB':>9s} {'write GB':>9s} {'calls':>6s}  kernel")
    tot = collections.Counter()
    for n in names:
        fg, wg, calls = 2 * f.get(n, (0, 0))[0], w.get(n, (0, 0))[0], f.get(n, w.get(n))[1]
        tot["f"] += fg; tot["w"] += wg
    for n in names[:top]:
        fg, wg, calls = 2 * f.get(n, (0, 0))[0], w.get(n, (0, 0))[0], f.get(n, w.get(n))[1]
        print(f"{fg:9.3f} {wg:9.3f} {calls:6.1f}  {short(n)}")
    print(f"{tot['f']:9.3f} {tot['w']:9.3f}         all kernels")


if __name__ == "__main__":
    main()
