"""HBM traffic of one train step from two rocprofv3 PMC passes (one counter per run: FETCH_SIZE, WRITE_SIZE), per kernel family.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_f -- python3 bench.py --steps 2 --warmup 2 --no-graph --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_w -- python3 bench.py --steps 2 --warmup 2 --no-graph --no-cpu-baseline
    python tools/pmc_traffic.py <fetch .db> <write .db> profiles/r4_hbm_traffic.json

The output records the date and bench.kernel_source_hash() (sha256 over csrc/, include/clite.h and the Python executors): bench.py reports
`roofline.traffic` only while that hash matches the tree it runs from, so a kernel change without new PMC passes yields null, not a stale number.

Both counters are reported in KiB; FETCH_SIZE is doubled (gfx950 tallies 64 B per 128-B request, MI355X_MICROARCH.md). Steps are counted by
the update kernel's dispatches (one per step); bench.py's per-launch roofline pass runs extra eager steps, which are steps like any other."""
import datetime
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_family(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name = ? group by kernel_name", (counter,)).fetchall()
    fam = {"igemm": [0.0, 0], "bn": [0.0, 0], "other": [0.0, 0]}
    steps = 0
    for name, val, n in rows:
        if "sgd_step_kernel" in name:
            steps = n
        key = "igemm" if ("igemm" in name or "splitk_finish" in name or "colstats_det" in name or "conv3x3_" in name or "wgrad_patch_reduce" in name) else "bn" if "bn_" in name else "other"
        fam[key][0] += val
        fam[key][1] += n
    return fam, steps


def main():
    fdb, wdb, out = sys.argv[1:4]
    f, sf = per_family(fdb, "FETCH_SIZE")
    w, sw = per_family(wdb, "WRITE_SIZE")
    kib = 1024.0
    import bench
    res = {
        "date": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"),
        "kernel_source_hash": bench.kernel_source_hash(),
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 2 --no-graph --no-cpu-baseline; tools/pmc_traffic.py",
        "correction": "FETCH_SIZE doubled (gfx950 tallies 64 B per 128-B request, MI355X_MICROARCH.md); WRITE_SIZE as reported; float-atomic traffic (weight gradients, split-K workspaces) is in WRITE_SIZE",
        "steps_in_pass": [sf, sw],
        "igemm_launches_per_step": round(f["igemm"][1] / sf, 1),
        "igemm_fetch_GB_per_step": round(2 * f["igemm"][0] * kib / sf / 1e9, 3),
        "igemm_write_GB_per_step": round(w["igemm"][0] * kib / sw / 1e9, 3),
        "bn_kernels_hbm_GB_per_step": round((2 * f["bn"][0] / sf + w["bn"][0] / sw) * kib / 1e9, 3),
        "other_kernels_hbm_GB_per_step": round((2 * f["other"][0] / sf + w["other"][0] / sw) * kib / 1e9, 3),
    }
    res["igemm_hbm_GB_per_step"] = round(res["igemm_fetch_GB_per_step"] + res["igemm_write_GB_per_step"], 3)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
