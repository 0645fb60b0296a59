#!/bin/bash
# The other configurations' bench lines (profiles/r3_bench_*.json), one gpurun call, one box — so that bf16 / fp8 pairs are same-box figures:
#   gpurun --timeout 1100 -- 'bash tools/collect_configs.sh'   then   cp gpurun_out/r3c/r3_bench_* profiles/
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c
mkdir -p $O
B="--no-cpu-baseline --steps 12 --warmup 3"
python bench.py $B --visual resnet101 --batch 256 > $O/r3_bench_rn101_b256.json 2> $O/err.log
python bench.py $B --visual resnet101 --batch 256 --fp8 > $O/r3_bench_rn101_b256_fp8.json 2>> $O/err.log
python bench.py $B --visual resnet101 --batch 256 > $O/r3_bench_rn101_b256_again.json 2>> $O/err.log
python bench.py $B --visual resnet101 --batch 256 --fp8 > $O/r3_bench_rn101_b256_fp8_again.json 2>> $O/err.log
python bench.py $B --fp8 > $O/r3_bench_fp8.json 2>> $O/err.log
python bench.py $B --f32 > $O/r3_bench_f32.json 2>> $O/err.log
python bench.py $B --loss infonce > $O/r3_bench_infonce.json 2>> $O/err.log
python bench.py $B --batch 256 > $O/r3_bench_rn50_b256.json 2>> $O/err.log
python bench.py $B --batch 1024 --steps 6 > $O/r3_bench_rn50_b1024.json 2>> $O/err.log
for f in $O/r3_bench_*.json; do python -c "import sys,json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],2), round(d['value']), d['dtype'][:12])"; done
