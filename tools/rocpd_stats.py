"""Turn a rocprofv3 rocpd database (`*_results.db`, ROCm 7.2 default output) into the per-kernel statistics CSV that
`rocprofv3 --kernel-trace --stats` used to write (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs), plus
a side-car JSON (third argument) with the implicit-GEMM family's in-step kernel time per step for bench.py's roofline.frac_in_step.
Usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db profiles/x_kernel_stats.csv [profiles/x_kernel_stats.json]"""
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
    if len(sys.argv) > 3:
        # side-car for bench.py's roofline.frac_in_step: the implicit-GEMM family's kernel time per step INSIDE the two-stream step (the same family
        # tools/pmc_traffic.py counts), steps = dispatches of the once-per-step weight-transpose kernel, and the hash of the tree it was taken on
        import datetime
        import json
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        fam = lambda n: "igemm" in n or "splitk_finish" in n or "conv3x3_" in n or "wgrad_patch" in n
        steps = sum(k for n, k, *_ in rows if "transpose_weights_kernel" in n) or 1
        ig = sum(s for n, k, s, *_ in rows if fam(n))
        bn = sum(s for n, k, s, *_ in rows if "bn_" in n and not fam(n))
        json.dump({"date": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"), "kernel_source_hash": bench.kernel_source_hash(), "steps": steps,
                   "igemm_ms_per_step": ig / steps / 1e6, "bn_kernels_ms_per_step": bn / steps / 1e6, "all_kernels_ms_per_step": tot / steps / 1e6,
                   "source": "rocprofv3 --kernel-trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-records; tools/rocpd_stats.py"},
                  open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
