// Layout probe for ds_read_b64_tr_b8 (gfx950): which LDS byte does byte j of lane l's 64-bit result come from, as a function of the addresses the
// lanes supply? LDS is filled with its own byte addresses (low byte in one pass, high byte in a second), every lane supplies address l * PITCH
// (+ an optional per-lane column offset), and the result is printed as source addresses. Build: hipcc --offload-arch=gfx950 -O2 tr8_probe.hip -o tr8_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef int i32x2 __attribute__((ext_vector_type(2)));

__global__ void probe(const int* addr, uint32_t* out, int pass) {
  __shared__ __attribute__((aligned(1024))) uint8_t lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = pass == 0 ? (uint8_t)(i & 0xff) : (uint8_t)(i >> 8);
  __syncthreads();
  const int l = threadIdx.x;
  i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)(lds + addr[l]));
  out[l * 2] = (uint32_t)v[0];
  out[l * 2 + 1] = (uint32_t)v[1];
}

int main() {
  int* daddr; uint32_t* dout;
  hipMalloc(&daddr, 64 * 4); hipMalloc(&dout, 128 * 4);
  for (int pat = 0; pat < 3; ++pat) {
    int addr[64];
    for (int l = 0; l < 64; ++l) addr[l] = pat == 0 ? l * 64 : pat == 1 ? (l & 15) * 64 + (l >> 4) * 8 : (l >> 3) * 128 + (l & 7) * 8;
    hipMemcpy(daddr, addr, sizeof(addr), hipMemcpyHostToDevice);
    uint32_t lo[128], hi[128];
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, daddr, dout, 0); hipMemcpy(lo, dout, sizeof(lo), hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, daddr, dout, 1); hipMemcpy(hi, dout, sizeof(hi), hipMemcpyDeviceToHost);
    printf("pattern %d (lane address: %s)\n", pat, pat == 0 ? "l * 64" : pat == 1 ? "(l & 15) * 64 + (l >> 4) * 8" : "(l >> 3) * 128 + (l & 7) * 8");
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d (addr %5d):", l, addr[l]);
      for (int j = 0; j < 8; ++j) {
        const int a = ((lo[l * 2 + (j >> 2)] >> (8 * (j & 3))) & 0xff) | (((hi[l * 2 + (j >> 2)] >> (8 * (j & 3))) & 0xff) << 8);
        // report as (supplying lane's address index, byte offset)
        int src_lane = -1, off = -1;
        for (int m = 0; m < 64; ++m) if (a >= addr[m] && a < addr[m] + 8) { src_lane = m; off = a - addr[m]; break; }
        printf(" %5d(L%d+%d)", a, src_lane, off);
      }
      printf("\n");
    }
  }
  return 0;
}
