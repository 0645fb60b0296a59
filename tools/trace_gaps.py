"""Idle time between consecutive kernels of each queue in one replayed step (rocprofv3 kernel trace, rocpd .db): how much of the
step's wall time on the critical stream is dispatch gaps rather than kernels.
    python tools/trace_gaps.py <results.db>"""
import sqlite3
import sys
from trace_neighbors import short


def main():
    c = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    q = "queue_id" if "queue_id" in cols else "0"
    rows = c.execute(f"select name, start, end, {q} from kernels order by start").fetchall()
    ends = [i for i, r in enumerate(rows) if "sumsq_partial_kernel" in r[0]]
    k = min(range(len(ends) - 1), key=lambda i: rows[ends[i + 1]][2] - rows[ends[i]][2])
    step = rows[ends[k] + 1:ends[k + 1] + 1]
    t0 = min(r[1] for r in step)
    print(f"{len(step)} kernels, step span {(max(r[2] for r in step) - t0) / 1e6:.3f} ms")
    for qid in sorted({r[3] for r in step}):
        seq = [r for r in step if r[3] == qid]
        busy = sum(r[2] - r[1] for r in seq)
        gaps = [(seq[i + 1][1] - seq[i][2], i) for i in range(len(seq) - 1)]
        small = sum(g for g, _ in gaps if 0 < g < 20000)
        big = [(g, i) for g, i in gaps if g >= 20000]
        print(f"queue {qid}: {len(seq)} kernels, busy {busy / 1e6:.3f} ms, span {(seq[-1][2] - seq[0][1]) / 1e6:.3f} ms, "
              f"gaps < 20 us: {small / 1e6:.3f} ms ({small / max(len(gaps), 1) / 1e3:.2f} us each), {len(big)} longer waits {sum(g for g, _ in big) / 1e6:.3f} ms")
        hist = {}
        for r in seq:
            d = r[2] - r[1]
            b = "<5us" if d < 5000 else "<10us" if d < 10000 else "<20us" if d < 20000 else "<50us" if d < 50000 else ">=50us"
            h = hist.setdefault(b, [0, 0])
            h[0] += 1
            h[1] += d
        print("   durations: " + ", ".join(f"{b}: {n} ({t / 1e6:.2f} ms)" for b, (n, t) in hist.items()))
        for g, i in sorted(big, reverse=True)[:6]:
            print(f"   wait {g / 1e3:7.1f} us after {short(seq[i][0])[:60]} before {short(seq[i + 1][0])[:60]}")


if __name__ == "__main__":
    main()
