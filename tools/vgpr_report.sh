#!/bin/bash
# Per-kernel register / LDS / occupancy report of the implicit-GEMM translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iclip-lite_amd/csrc -Wno-unused-value -Rpass-analysis=kernel-resource-usage -c clip-lite_amd/csrc/gemm.hip -o /tmp/gemm_ru.o 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs:|Occupancy|LDS Size" | sed -e 's/.*remark: [^ ]* *//' -e 's/ \[-Rpass.*//' | paste - - - - - | sed -e 's/Function Name: _ZN5clite//' | awk '{print}' | cut -c1-260
