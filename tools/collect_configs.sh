#!/bin/bash
# The other configurations' bench lines (profiles/r5_bench_*.json; round 5: + the fp8 step without the fp8 weight gradients), one gpurun call, one box — so that bf16 / fp8 pairs are same-box figures:
#   gpurun --timeout 1100 -- 'bash tools/collect_configs.sh'   then   cp gpurun_out/r5c/r5_bench_* profiles/
# *_fp8_nonscaled*: the same step on build/varf8's library (tools/build_patch_variants.sh: the fp8 forward on v_mfma_f32_32x32x16_fp8_fp8, round 3's
# form) — what the block-scaled instruction is worth inside the step.
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c
mkdir -p $O
B="--no-cpu-baseline --no-side-records --steps 12 --warmup 3"
V=build/varf8/libclite_hip_var.so
python bench.py $B --visual resnet101 --batch 256 > $O/r5_bench_rn101_b256.json 2> $O/err.log
python bench.py $B --visual resnet101 --batch 256 --fp8 > $O/r5_bench_rn101_b256_fp8.json 2>> $O/err.log
[ -f $V ] && CLITE_HIP_LIB=$V python bench.py $B --visual resnet101 --batch 256 --fp8 > $O/r5_bench_rn101_b256_fp8_nonscaled.json 2>> $O/err.log
python bench.py $B --visual resnet101 --batch 256 --fp8 --no-fp8-wgrad > $O/r5_bench_rn101_b256_fp8_bf16_wgrad.json 2>> $O/err.log
python bench.py $B --visual resnet101 --batch 256 --fp8 --no-fp8-text --no-fp8-dgrad > $O/r5_bench_rn101_b256_fp8_image_only.json 2>> $O/err.log
python bench.py $B --visual resnet101 --batch 256 --fp8 --no-fp8-dgrad > $O/r5_bench_rn101_b256_fp8_forward_only.json 2>> $O/err.log
python bench.py $B --visual resnet101 --batch 256 > $O/r5_bench_rn101_b256_again.json 2>> $O/err.log
python bench.py $B --visual resnet101 --batch 256 --fp8 > $O/r5_bench_rn101_b256_fp8_again.json 2>> $O/err.log
python bench.py $B > $O/r5_bench_rn50_b128.json 2>> $O/err.log
python bench.py $B --fp8 > $O/r5_bench_fp8.json 2>> $O/err.log
[ -f $V ] && CLITE_HIP_LIB=$V python bench.py $B --fp8 > $O/r5_bench_fp8_nonscaled.json 2>> $O/err.log
python bench.py $B --fp8 --no-fp8-text --no-fp8-dgrad > $O/r5_bench_fp8_image_only.json 2>> $O/err.log
python bench.py $B --fp8 --no-fp8-dgrad > $O/r5_bench_fp8_forward_only.json 2>> $O/err.log
python bench.py $B --loss infonce > $O/r5_bench_infonce.json 2>> $O/err.log
python bench.py $B --batch 256 > $O/r5_bench_rn50_b256.json 2>> $O/err.log
for f in $O/r5_bench_*.json; do python -c "import sys,json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],2), round(d['value']), d['dtype'][:12])"; done
