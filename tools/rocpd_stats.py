"""Turn a rocprofv3 rocpd database (`*_results.db`, ROCm 7.2 default output) into the per-kernel statistics CSV that
`rocprofv3 --kernel-trace --stats` used to write (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs), plus
an optional per-dispatch listing. Usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db profiles/x_kernel_stats.csv"""
import csv
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
    tot = float(sum(r[2] for r in rows)) or 1.0
    with open(out, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, k, s, a, mn, mx in rows:
            w.writerow([n, k, s, round(a, 3), round(100.0 * s / tot, 4), mn, mx])
    print(f"{len(rows)} kernels, {tot / 1e6:.3f} ms of kernel time -> {out}")


if __name__ == "__main__":
    main()
