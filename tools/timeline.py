"""Stream timeline of the captured train step from a rocprofv3 kernel trace (rocpd .db): per steady-state step, how long each HIP stream
had a kernel running, how long both / neither did, and the largest idle gaps with the kernels around them. Usage:
    rocprofv3 --kernel-trace -d gpurun_out/tl -- python3 bench.py --steps 6 --warmup 4 --no-cpu-baseline
    python tools/timeline.py gpurun_out/tl/*/*_results.db"""
import sqlite3
import sys


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def length(iv):
    return sum(b - a for a, b in iv)


def intersect(x, y):
    i = j = 0
    out = []
    while i < len(x) and j < len(y):
        a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        if a < b:
            out.append([a, b])
        if x[i][1] < y[j][1]:
            i += 1
        else:
            j += 1
    return out


def main():
    c = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    sid = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)      # (stream ids need --hip-trace; HW queues do not)
    rows = c.execute(f"select name, start, end, {sid or 0} from kernels order by start").fetchall()
    ends = [r[2] for r in rows if "sumsq_partial_kernel" in r[0]]
    if len(ends) < 4:
        print("not enough steps in the trace")
        return
    # three consecutive steps of minimal total duration = the replayed (steady-state) ones, not the eager warm-up / per-launch timing steps
    k = min(range(len(ends) - 3), key=lambda i: ends[i + 3] - ends[i])
    t0, t1 = ends[k], ends[k + 3]
    sel = [r for r in rows if r[1] >= t0 and r[2] <= t1]
    streams = {}
    for n, a, b, s in sel:
        streams.setdefault(s, []).append((a, b))
    wall = (t1 - t0) / 3e6
    print(f"3 steps, {wall:.3f} ms per step, {len(sel) // 3} kernels per step, streams: {len(streams)}")
    un = {s: union(iv) for s, iv in streams.items()}
    order = sorted(un, key=lambda s: -length(un[s]))
    for s in order:
        print(f"  stream {s}: busy {length(un[s]) / 3e6:7.3f} ms/step in {len(streams[s]) // 3} kernels")
    if len(order) >= 2:
        both = intersect(un[order[0]], un[order[1]])
        anyb = union([tuple(x) for s in order for x in un[s]])
        print(f"  both top streams busy {length(both) / 3e6:.3f} ms/step; no kernel running at all {wall - length(anyb) / 3e6:.3f} ms/step")
    # largest gaps on the busiest stream
    main_iv = sorted(streams[order[0]])
    names = {(a, b): n for n, a, b, s in sel if s == order[0]}
    gaps = []
    for (a0, b0), (a1, b1) in zip(main_iv, main_iv[1:]):
        if a1 - b0 > 20000:
            gaps.append((a1 - b0, names[(a0, b0)][:60], names[(a1, b1)][:60]))
    print(f"  gaps > 20 us on the busiest stream: {len(gaps) // 3} per step, {sum(g[0] for g in gaps) / 3e6:.3f} ms/step")
    for g in sorted(gaps, reverse=True)[:12]:
        print(f"    {g[0] / 1e3:8.1f} us between {g[1]} -> {g[2]}")
    small = sum(a1 - b0 for (a0, b0), (a1, b1) in zip(main_iv, main_iv[1:]) if 0 < a1 - b0 <= 20000)
    print(f"  gaps <= 20 us on the busiest stream: {small / 3e6:.3f} ms/step")
    # Gantt of the middle step: per queue, maximal runs of kernels separated by < 100 us
    s0, s1 = ends[k + 1], ends[k + 2]
    print(f"  one step ({(s1 - s0) / 1e6:.3f} ms), runs of kernels (gap < 100 us) per queue; times in ms from the step start:")
    short = lambda n: n.replace("_ZN12_GLOBAL__N_1", "").replace("_ZN5clite", "").replace("void ", "")[:44]
    for q in order:
        ks = sorted((a, b, n) for n, a, b, s in sel if s == q and a >= s0 and b <= s1)
        runs, cur = [], None
        for a, b, n in ks:
            if cur is not None and a - cur[1] < 100000:
                cur[1], cur[3], cur[4], cur[5] = b, n, cur[4] + 1, cur[5] + (b - a)
            else:
                cur = [a, b, n, n, 1, b - a]
                runs.append(cur)
        for a, b, n0, n1, cnt, busy in runs:
            print(f"    q{q} {(a - s0) / 1e6:7.3f} - {(b - s0) / 1e6:7.3f}  ({cnt:3d} kernels, busy {busy / 1e6:6.3f})  {short(n0)} ... {short(n1)}")


if __name__ == "__main__":
    main()
