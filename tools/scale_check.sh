#!/bin/bash
# Pre-flight of the driver's multi-GPU bench on a ONE-GPU box: the full-size data-parallel step — ResNet-50 + BERT-base,
# per-rank batch 128, captured per-phase hipGraphs with the gradient-region exchanges between them — as two ranks that share cuda:0 over gloo
# (`--single-device`; RCCL needs one GPU per rank), for both exchange forms. Asserts what the real run must also show: one JSON line, n_gpus 2,
# the collective pre-flight counted two ranks, hipGraph replay, replicas bit-identical after the timed steps, a finite loss.
#   gpurun --timeout 900 -- 'bash tools/scale_check.sh'          (two processes on the card: within the box's process guard)
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
for ex in allreduce mesh; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29547 bench.py --gpus 2 --steps 5 --warmup 3 \
    --backend gloo --single-device --no-cpu-baseline --exchange $ex > gpurun_out/scale_check_$ex.json 2> gpurun_out/scale_check_$ex.err
  python - "$ex" <<'PY'
import json, math, sys
ex = sys.argv[1]
lines = [l for l in open(f"gpurun_out/scale_check_{ex}.json") if l.strip()]
assert len(lines) == 1, lines
r = json.loads(lines[0])
assert r["n_gpus"] == 2 and r["rccl_ranks"] == 2 and r["config"]["global_batch"] == 256, r
assert r["launch"] == "hipGraph replay" and r["replicas_identical"] is True and math.isfinite(r["loss"]), r
print(f"scale_check {ex}: ok — {r['ms_per_step']:.2f} ms/step with two ranks on one GPU (timing meaningless), replicas identical, loss {r['loss']:.4f}")
PY
done
