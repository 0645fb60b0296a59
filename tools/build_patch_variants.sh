#!/bin/bash
# Ablation variants of the product library that differ in csrc/conv_patch.hip only: build/var{1,2,4,3}/libclite_hip_var.so (run after `make hip`).
set -e
cd "$(dirname "$0")/.."
for v in 1 2 4 3; do
  mkdir -p build/var$v
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iclip-lite_amd/csrc -Wno-unused-value -DCLITE_PATCH_ABLATE=$v -c clip-lite_amd/csrc/conv_patch.hip -o build/var$v/conv_patch.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/var$v/libclite_hip_var.so build/var$v/conv_patch.o $(ls build/hip/*.o | grep -v conv_patch.o)
done
