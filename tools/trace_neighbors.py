"""Which library kernels surround the runtime's own copy / fill kernels in one replayed step (rocprofv3 kernel trace, rocpd .db):
    python tools/trace_neighbors.py <results.db> [pattern ...]     default patterns: copyBuffer at::native"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+|_ZN5clite\d+|\(anonymous namespace\)::|clite::|void ", "", n)
    return n[:70]


def main():
    c = sqlite3.connect(sys.argv[1])
    pats = sys.argv[2:] or ["copyBuffer", "at::native"]
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    q = "queue_id" if "queue_id" in cols else "0"
    rows = c.execute(f"select name, start, end, {q} from kernels order by start").fetchall()
    ends = [i for i, r in enumerate(rows) if "sumsq_partial_kernel" in r[0]]
    k = min(range(len(ends) - 1), key=lambda i: rows[ends[i + 1]][2] - rows[ends[i]][2])       # the shortest step = a replayed one
    step = rows[ends[k] + 1:ends[k + 1] + 1]
    print(f"{len(step)} kernels in the step")
    for qid in sorted({r[3] for r in step}):
        seq = [r for r in step if r[3] == qid]
        for i, r in enumerate(seq):
            if any(p in r[0] for p in pats):
                prev = short(seq[i - 1][0]) if i else "-"
                nxt = short(seq[i + 1][0]) if i + 1 < len(seq) else "-"
                print(f"q{qid} {(r[2] - r[1]) / 1e3:6.1f} us {short(r[0])[:40]:40s} after {prev[:50]:50s} before {nxt[:50]}")


if __name__ == "__main__":
    main()
