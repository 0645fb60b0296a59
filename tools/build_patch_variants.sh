#!/bin/bash
# Ablation variants of the product library that differ in csrc/conv_patch.hip only: build/var{1,2,4,3}/libclite_hip_var.so (run after `make hip`).
set -e
cd "$(dirname "$0")/.."
mkdir -p build/varnt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iclip-lite_amd/csrc -Wno-unused-value -DCLITE_PATCH_NT=0 -c clip-lite_amd/csrc/conv_patch.hip -o build/varnt/conv_patch.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/varnt/libclite_hip_var.so build/varnt/conv_patch.o $(ls build/hip/*.o | grep -v conv_patch.o)
# the round-3 kernel selection on this round's sources: no patch-resident kernels, no compile-time BERT epilogue forms (same-box A/B of the step)
mkdir -p build/varr3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iclip-lite_amd/csrc -Wno-unused-value -DCLITE_NO_PATCH -c clip-lite_amd/csrc/conv_patch.hip -o build/varr3/conv_patch.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iclip-lite_amd/csrc -Wno-unused-value -DCLITE_NO_BERT_FORMS -c clip-lite_amd/csrc/gemm_wide.hip -o build/varr3/gemm_wide.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/varr3/libclite_hip_var.so build/varr3/conv_patch.o build/varr3/gemm_wide.o $(ls build/hip/*.o | grep -v -e conv_patch.o -e gemm_wide.o)
mkdir -p build/varat
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iclip-lite_amd/csrc -Wno-unused-value -DCLITE_GROUP_ATOMIC_ALWAYS -c clip-lite_amd/csrc/gemm_group.hip -o build/varat/gemm_group.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/varat/libclite_hip_var.so build/varat/gemm_group.o $(ls build/hip/*.o | grep -v gemm_group.o)
for v in ${PATCH_VARIANTS:-1 2 4 3}; do
  mkdir -p build/var$v
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iclip-lite_amd/csrc -Wno-unused-value -DCLITE_PATCH_ABLATE=$v -c clip-lite_amd/csrc/conv_patch.hip -o build/var$v/conv_patch.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/var$v/libclite_hip_var.so build/var$v/conv_patch.o $(ls build/hip/*.o | grep -v conv_patch.o)
done
# the fp8 forward on the non-scaled v_mfma_f32_32x32x16_fp8_fp8 (round 3's form): same-box A/B of the block-scaled instruction
mkdir -p build/varf8
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iclip-lite_amd/csrc -Wno-unused-value -DCLITE_FP8_SCALED=0 -c clip-lite_amd/csrc/fp8_ops.hip -o build/varf8/fp8_ops.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/varf8/libclite_hip_var.so build/varf8/fp8_ops.o $(ls build/hip/*.o | grep -v fp8_ops.o)
# the HBM-bound 1 x 1 forward convolutions as one tile per workgroup (before the row-range persistent FORM 4): same-box A/B, tools/probe_fwd1x1.py
mkdir -p build/varnpf
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iclip-lite_amd/csrc -Wno-unused-value -DCLITE_NO_PERSIST_FWD -c clip-lite_amd/csrc/gemm.hip -o build/varnpf/gemm.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/varnpf/libclite_hip_var.so build/varnpf/gemm.o $(ls build/hip/*.o | grep -v "/gemm.o")
# the stem's patch-resident weight gradient in its 8-wave double-buffered form (tools/probe_stem.py)
mkdir -p build/varsw8
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iclip-lite_amd/csrc -Wno-unused-value -DCLITE_STEM_WGRAD_NW=8 -c clip-lite_amd/csrc/conv_patch.hip -o build/varsw8/conv_patch.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/varsw8/libclite_hip_var.so build/varsw8/conv_patch.o $(ls build/hip/*.o | grep -v conv_patch.o)
