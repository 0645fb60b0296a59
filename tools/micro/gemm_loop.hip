// Microbenchmark: the implicit-GEMM main loop's memory side in isolation. One 256-thread workgroup per output tile streams K tiles of A
// (128 rows x 64 B) and B (128 rows x 64 B) through a 3-stage LDS ring with `buffer_load ... lds`, counted vmcnt and one s_barrier per
// K tile, exactly like igemm_dma_kernel, but with no LDS reads / MFMAs and precomputed addresses. Variants: barrier on/off.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ void dma16(rsrc_t r, uint32_t off, void* lds) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool BARRIER, int NSTAGE, int ROT = 0>
__global__ __launch_bounds__(256) void loop_kernel(const char* A, uint32_t abytes, const char* B, uint32_t bbytes, int pitch, int tiles_n, int ktiles, int* sink, int wrap) {
  __shared__ __attribute__((aligned(1024))) char lds[NSTAGE * 16384];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwg = gridDim.x, xcd = blockIdx.x & 7, xq = nwg >> 3, xr = nwg & 7;
  const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
  // tiles_n < 0: every workgroup streams private rows (no panel shared between workgroups)
  const int tm = tiles_n > 0 ? wg / tiles_n : wg % 240, tn = tiles_n > 0 ? wg % tiles_n : (wg * 7 + 3) % 240;
  rsrc_t ra = make_rsrc(A, abytes), rb = make_rsrc(B, bbytes);
  uint32_t offa[2], offb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int row = (wave * 2 + j) * 16 + (lane >> 2);
    offa[j] = (uint32_t)((tm * 128 + row) * (size_t)pitch + (lane & 3) * 16);
    offb[j] = (uint32_t)((tn * 128 + row) * (size_t)pitch + (lane & 3) * 16);
  }
  // ROT: workgroup w starts its K loop at tile (w * ROT) % ktiles and wraps, so concurrent workgroups are in different 64-B columns
  // (different L2 channels) of a power-of-two-pitch operand
  int kcur = ROT ? (int)(((long)wg * ROT) % ktiles) : 0;
  auto issue = [&](int buf) {
    char* base = lds + buf * 16384;
    const uint32_t kb = (uint32_t)kcur * 64;
#pragma unroll
    for (int j = 0; j < 2; ++j) dma16(ra, offa[j] + kb, base + (wave * 2 + j) * 1024);
#pragma unroll
    for (int j = 0; j < 2; ++j) dma16(rb, offb[j] + kb, base + 8192 + (wave * 2 + j) * 1024);
    if (++kcur == wrap) kcur = 0;
  };
#pragma unroll
  for (int p = 0; p < NSTAGE - 1; ++p) issue(p);
  int buf = 0;
  for (int t = 0; t < ktiles; ++t) {
    const int after = ktiles - 1 - t;
    if (NSTAGE >= 4 && after >= 2) wait_vm<8>();
    else if (after >= 1) wait_vm<4>();
    else wait_vm<0>();
    if (BARRIER) __builtin_amdgcn_s_barrier();
    if (t + NSTAGE - 1 < ktiles) { int nb = buf + NSTAGE - 1; if (nb >= NSTAGE) nb -= NSTAGE; issue(nb); }
    if (++buf == NSTAGE) buf = 0;
  }
  __syncthreads();
  if (threadIdx.x == 0 && lds[123] == 77) sink[0] = 1;
}

template <bool BARRIER, int NSTAGE, int ROT = 0>
void run(const char* name, char* A, char* B, int M, int N, int K, int* sink, int pad = 0, bool priv = false, int loops = 1) {
  int tiles_n = N / 128, tiles = (M / 128) * tiles_n, ktiles = K / 32, pitch = K * 2 + pad;
  if (priv) tiles_n = -tiles_n;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i)
      hipLaunchKernelGGL((loop_kernel<BARRIER, NSTAGE, ROT>), dim3(tiles), dim3(256), 0, 0, A, priv ? (256u << 20) : (uint32_t)((size_t)M * pitch), B, priv ? (256u << 20) : (uint32_t)((size_t)N * pitch), pitch, tiles_n, ktiles * loops, sink, ktiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  ktiles *= loops;
  double gb = (double)tiles * ktiles * 16384 / 1e9;
  printf("%-28s M=%5d N=%5d K=%5d tiles %4d: %7.1f us  %7.1f GB/s  %6.3f us per K tile per workgroup-round\n", name, M, N, K, tiles, ms * 1e3, gb / (ms * 1e-3),
         ms * 1e3 / ktiles / ((tiles + 767) / 768));
}

int main() {
  char *A, *B; int* sink;
  hipMalloc(&A, 256u << 20); hipMalloc(&B, 256u << 20); hipMemset(A, 1, 256u << 20); hipMemset(B, 1, 256u << 20); hipMalloc(&sink, 4);
  int shapes[][3] = {{3840, 768, 768}, {3840, 768, 256}, {1920, 768, 768}, {7680, 1536, 256}, {3840, 768, 3072}, {3840, 3072, 768}, {3840, 2304, 768}, {4096, 4096, 4096}, {25088, 256, 2304}};
  for (auto& s : shapes) {
    run<true, 3>("barrier, 3 stages", A, B, s[0], s[1], s[2], sink);
    run<false, 3>("no barrier, 3 stages", A, B, s[0], s[1], s[2], sink);
    run<true, 4>("barrier, 4 stages", A, B, s[0], s[1], s[2], sink);
    run<true, 3>("3 stages, K swept 8x", A, B, s[0], s[1], s[2], sink, 0, false, 8);
    run<true, 4>("4 stages, K swept 8x", A, B, s[0], s[1], s[2], sink, 0, false, 8);
    run<true, 3>("3 stages, private rows", A, B, s[0], s[1], s[2], sink, 0, true);
    run<true, 3>("3 stages, private, +128", A, B, s[0], s[1], s[2], sink, 128, true);
  }
  return 0;
}
