"""Matrix-pipe utilisation of one train step by kernel family from two rocprofv3 PMC passes over `bench.py --no-graph` (separate runs):
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES -d A -o a -- python3 bench.py --steps 2 --warmup 2 --no-graph --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE          -d B -o b -- python3 bench.py --steps 2 --warmup 2 --no-graph --no-cpu-baseline
    python tools/pmc_mfma.py A/a_results.db B/b_results.db
SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs (MI355X_MICROARCH.md), so the fraction of SIMD
cycles with the matrix pipe busy while a kernel runs is MFMA_BUSY / (1024 * GUI_ACTIVE / 8)."""
import collections
import sqlite3
import sys


def table(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name = ? group by kernel_name", (counter,)).fetchall()
    steps = max((n for name, _, n in rows if "sgd_step_kernel" in name), default=1)
    return {name: val / steps for name, val, n in rows}


def family(n):
    if "conv3x3_" in n:
        return "patch-resident 3 x 3 kernels"
    if "igemm_group" in n:
        return "igemm grouped weight gradients"
    if "igemm_wide" in n:
        return "igemm wide (8-wave) kernels"
    if "igemm" in n:
        return "igemm narrow (4-wave) kernels"
    if "attention" in n:
        return "attention (MFMA)"
    return "everything else (no MFMA)"


def main():
    busy, act = table(sys.argv[1], "SQ_VALU_MFMA_BUSY_CYCLES"), table(sys.argv[2], "GRBM_GUI_ACTIVE")
    fam = collections.defaultdict(lambda: [0.0, 0.0])
    for n in set(busy) | set(act):
        f = fam[family(n)]
        f[0] += busy.get(n, 0.0)
        f[1] += act.get(n, 0.0)
    print(f"{'kernel family':40s} {'ms/step (GUI active)':>21s} {'matrix pipe busy':>17s}")
    tb = ta = 0.0
    for k, (b, a) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        tb += b; ta += a
        print(f"{k:40s} {a / 8 / 2.4e6:21.3f} {b / (1024 * a / 8) if a else 0:17.3f}")
    print(f"{'whole step (kernels back to back)':40s} {ta / 8 / 2.4e6:21.3f} {tb / (1024 * ta / 8):17.3f}")


if __name__ == "__main__":
    main()
