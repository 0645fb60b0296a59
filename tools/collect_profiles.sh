#!/bin/bash
# One gpurun call that regenerates what profiles/r5_* is made from (run from the repository root on the GPU box):
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh'   then   cp gpurun_out/r5p/r5_* profiles/
# PMC passes are separate runs with --kernel-trace only (one counter each), as MI355X_MICROARCH.md prescribes. The traffic pass runs first and
# its result and the kernel trace's side-car are put under profiles/ of the box's copy, so that the bench line taken afterwards carries roofline.traffic,
# roofline.hbm and roofline.frac_in_step for this very tree.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5p
mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o f -- python3 bench.py --steps 2 --warmup 2 --no-graph --no-cpu-baseline > $O/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o w -- python3 bench.py --steps 2 --warmup 2 --no-graph --no-cpu-baseline > $O/pmc_w.log 2>&1
python tools/pmc_traffic.py $O/pmc_f/f_results.db $O/pmc_w/w_results.db $O/r5_hbm_traffic.json
python tools/pmc_by_kernel.py $O/pmc_f/f_results.db $O/pmc_w/w_results.db 60 > $O/r5_pmc_by_kernel.txt
cp $O/r5_hbm_traffic.json profiles/r5_hbm_traffic.json
rocprofv3 --kernel-trace -d $O/kt -o kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-records > $O/kt.log 2>&1
python tools/rocpd_stats.py $O/kt/kt_results.db $O/r5_kernel_stats.csv $O/r5_kernel_stats.json
cp $O/r5_kernel_stats.json profiles/r5_kernel_stats.json
python bench.py > $O/r5_bench.json 2> $O/bench.err
python tools/diag_launch.py > $O/r5_timeline.txt 2>&1
python tools/layer_profile.py > $O/r5_layers.txt 2> $O/layers.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES -d $O/pmc_m -o m -- python3 bench.py --steps 2 --warmup 2 --no-graph --no-cpu-baseline > $O/pmc_m.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/pmc_g -o g -- python3 bench.py --steps 2 --warmup 2 --no-graph --no-cpu-baseline > $O/pmc_g.log 2>&1
python tools/pmc_mfma.py $O/pmc_m/m_results.db $O/pmc_g/g_results.db > $O/r5_mfma_busy.txt
rm -rf $O/pmc_f $O/pmc_w $O/pmc_m $O/pmc_g $O/kt
