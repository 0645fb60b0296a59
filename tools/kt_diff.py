"""Per-kernel time per step of two rocprofv3 kernel traces (rocpd .db) side by side: python tools/kt_diff.py a.db b.db [steps_a steps_b]
Only the kernels whose per-step time differs by more than 10 us are listed."""
import sqlite3
import sys


def load(db, steps):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start) from kernels group by name").fetchall()
    return {n: (k / steps, s / steps / 1e3) for n, k, s in rows}


def main():
    sa = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    sb = float(sys.argv[4]) if len(sys.argv) > 4 else sa
    a, b = load(sys.argv[1], sa), load(sys.argv[2], sb)
    short = lambda n: n.replace("_ZN12_GLOBAL__N_1", "").replace("_ZN5clite", "").replace("void ", "").replace("(anonymous namespace)::", "")[:110]
    tot_a = tot_b = 0.0
    out = []
    for n in sorted(set(a) | set(b)):
        ka, ta = a.get(n, (0, 0.0))
        kb, tb = b.get(n, (0, 0.0))
        tot_a += ta
        tot_b += tb
        if abs(ta - tb) > 10:
            out.append((tb - ta, ka, ta, kb, tb, short(n)))
    for d, ka, ta, kb, tb, n in sorted(out):
        print(f"{d:+9.1f} us  {ka:6.1f} x {ta:8.1f} -> {kb:6.1f} x {tb:8.1f}  {n}")
    print(f"total kernel time per step: {tot_a / 1e3:.3f} -> {tot_b / 1e3:.3f} ms")


if __name__ == "__main__":
    main()
