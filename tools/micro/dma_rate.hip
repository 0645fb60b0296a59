// Microbenchmark: per-CU throughput of LDS-DMA (buffer_load_dwordx4 ... lds) from an L2-resident working set, by access shape
// (rows x bytes per wave-instruction), in-flight depth and waves per CU. Build: hipcc --offload-arch=gfx950 -O3 dma_rate.hip -o dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ void dma16(rsrc_t r, uint32_t off, void* lds) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Each wave streams `iters` groups of DEPTH instructions. Instruction covers ROWS rows x (1024/ROWS) bytes; rows are `pitch` bytes apart.
// Working set per workgroup: a private slab of `slab` bytes (re-read cyclically) so everything stays in L2 after the first touch.
template <int ROWS, int DEPTH>
__global__ __launch_bounds__(256) void dma_kernel(const char* src, uint32_t bytes, int pitch, int slab, int iters, int* sink) {
  __shared__ __attribute__((aligned(1024))) char lds[4 * DEPTH * 2 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  rsrc_t rs = make_rsrc(src, bytes);
  constexpr int BPR = 1024 / ROWS;            // bytes per row per instruction
  constexpr int LPR = BPR / 16;               // lanes per row
  const int row = lane / LPR, col = (lane % LPR) * 16;
  uint32_t base = (uint32_t)((blockIdx.x % (bytes / slab)) * (size_t)slab);
  uint32_t off = base + (uint32_t)((wave * ROWS + row) * pitch + col);
  const uint32_t step = (uint32_t)(4 * ROWS * pitch);      // all 4 waves advance together over the slab
  uint32_t pos = 0;
  // prologue: one group in flight
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) { dma16(rs, off + pos, lds + ((wave * 2 + 0) * DEPTH + d) * 1024); pos += step; if (pos + step > (uint32_t)slab) pos = 0; }
  for (int it = 1; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { dma16(rs, off + pos, lds + ((wave * 2 + (it & 1)) * DEPTH + d) * 1024); pos += step; if (pos + step > (uint32_t)slab) pos = 0; }
    wait_vm<DEPTH>();                          // the older group has landed
  }
  wait_vm<0>();
  __syncthreads();
  if (threadIdx.x == 0 && lds[123] == 77) sink[0] = 1;
}

template <int ROWS, int DEPTH>
float run(const char* src, uint32_t bytes, int pitch, int slab, int wgs, int iters, int* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((dma_kernel<ROWS, DEPTH>), dim3(wgs), dim3(256), 0, 0, src, bytes, pitch, slab, iters, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((dma_kernel<ROWS, DEPTH>), dim3(wgs), dim3(256), 0, 0, src, bytes, pitch, slab, iters, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double gb = (double)wgs * 4 * iters * DEPTH * 1024 / 1e9;
  printf("rows/instr %2d (%4d B/row) depth %d KB/wave %2d  wgs %4d: %8.1f us  %7.1f GB/s chip  %6.1f GB/s per CU\n", ROWS, 1024 / ROWS, DEPTH, DEPTH,
         wgs, ms * 1e3, gb / (ms * 1e-3), gb / (ms * 1e-3) / 256);
  return ms;
}

int main() {
  uint32_t bytes = 64u << 20;                  // 64 MB source; each workgroup re-reads a 128 KB slab -> L2 resident (256-768 slabs)
  char* src; int* sink;
  hipMalloc(&src, bytes); hipMemset(src, 1, bytes); hipMalloc(&sink, 4);
  const int iters = 400;
  for (int wgs : {256, 512, 768}) {
    // KC-32: 16 rows x 64 B, row pitch 1536 B (K = 768 bf16)
    run<16, 2>(src, bytes, 1536, 128 << 10, wgs, iters, sink);
    run<16, 4>(src, bytes, 1536, 128 << 10, wgs, iters, sink);
    run<16, 8>(src, bytes, 1536, 128 << 10, wgs, iters, sink);
    // KC-64: 8 rows x 128 B
    run<8, 4>(src, bytes, 1536, 128 << 10, wgs, iters, sink);
    run<8, 8>(src, bytes, 1536, 128 << 10, wgs, iters, sink);
    // XC-128: 4 rows x 256 B
    run<4, 4>(src, bytes, 1536, 128 << 10, wgs, iters, sink);
    run<4, 8>(src, bytes, 1536, 128 << 10, wgs, iters, sink);
    // fully contiguous 1 KB
    run<1, 4>(src, bytes, 1024, 128 << 10, wgs, iters, sink);
    run<1, 8>(src, bytes, 1024, 128 << 10, wgs, iters, sink);
  }
  return 0;
}
