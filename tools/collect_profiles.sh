cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r2p && \
python bench.py > gpurun_out/r2p/bench.json 2> gpurun_out/r2p/bench.err && \
python tools/diag_launch.py > gpurun_out/r2p/timeline.txt 2>&1 && \
rocprofv3 --kernel-trace -d gpurun_out/r2p/kt -o kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2p/kt.log 2>&1 && \
python tools/rocpd_stats.py gpurun_out/r2p/kt/kt_results.db gpurun_out/r2p/kernel_stats.csv && \
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r2p/pmc_f -o f -- python3 bench.py --steps 2 --warmup 2 --no-graph --no-cpu-baseline > gpurun_out/r2p/pmc_f.log 2>&1 && \
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r2p/pmc_w -o w -- python3 bench.py --steps 2 --warmup 2 --no-graph --no-cpu-baseline > gpurun_out/r2p/pmc_w.log 2>&1 && \
python tools/pmc_traffic.py gpurun_out/r2p/pmc_f/f_results.db gpurun_out/r2p/pmc_w/w_results.db gpurun_out/r2p/r2_hbm_traffic.json && rm -rf gpurun_out/r2p/pmc_f gpurun_out/r2p/pmc_w gpurun_out/r2p/kt
