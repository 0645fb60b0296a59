"""Every kernel of one replayed step between the first kernel matching FROM and the first (later) one matching TO, all queues, with start
offsets and durations (rocprofv3 kernel trace, rocpd .db) — e.g. the loss heads between the two average-pool kernels:
    python tools/trace_window.py <results.db> avgpool_fwd avgpool_bwd
TO may be a number: that many kernels from FROM on."""
import sqlite3
import sys
from trace_neighbors import short


def main():
    c = sqlite3.connect(sys.argv[1])
    a, b = sys.argv[2], sys.argv[3]
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    q = "queue_id" if "queue_id" in cols else "0"
    rows = c.execute(f"select name, start, end, {q} from kernels order by start").fetchall()
    ends = [i for i, r in enumerate(rows) if "sumsq_partial_kernel" in r[0]]
    k = min(range(len(ends) - 1), key=lambda i: rows[ends[i + 1]][2] - rows[ends[i]][2])
    step = rows[ends[k] + 1:ends[k + 1] + 1]
    i0 = next(i for i, r in enumerate(step) if a in r[0])
    i1 = min(i0 + int(b), len(step) - 1) if b.isdigit() else next(i for i, r in enumerate(step) if i > i0 and b in r[0])
    t0 = step[i0][1]
    for r in step[i0:i1 + 1]:
        print(f"q{r[3]} {(r[1] - t0) / 1e3:8.1f} us  +{(r[2] - r[1]) / 1e3:6.1f} us  {short(r[0])[:110]}")


if __name__ == "__main__":
    main()
