// Microbenchmark (round 5): does the WEIGHT operand of a 128 x 128 x 64 tile cost less when it goes from L2 straight into registers (global_load_dwordx4 in
// the MFMA fragment pattern: lane l reads 16 B = 8 k of column l & 31) than through the LDS-DMA ring beside the activation operand? Memory side only (no
// LDS reads, no MFMAs), 8 waves, 128-byte K rows, 3 stages, counted vmcnt, one barrier per K tile — csrc/igemm_wide.h's W128 loop.
//   mode 0: A and B by `buffer_load ... lds` (the product)      mode 1: A by LDS-DMA, B by register loads two tiles ahead      mode 2: A alone      mode 3: B alone
// Each shape twice: dense rows (pitch = 2 K bytes) and rows padded by 128 bytes (is the 24 x 256-byte pitch of K = 3072 a channel-aliasing problem?)
// Build on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/micro/breg_loop.hip -o tools/micro/breg_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ void dma16(rsrc_t r, uint32_t off, void* lds) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ u32x4 gload(const char* p) { u32x4 r; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p)); return r; }
__device__ __forceinline__ void use(u32x4& v, uint32_t& acc) { asm volatile("" : "+v"(v)); acc ^= v[0] ^ v[3]; }

template <int MODE>
__global__ __launch_bounds__(512) void loop_kernel(const char* A, uint32_t abytes, const char* B, uint32_t bbytes, int pitch, int tiles_n, int ktiles, int* sink) {
  constexpr int BM = 128, BN = 128, RB = 128, NW = 8, NSTAGE = 3;
  constexpr int STAGE = (BM + BN) * RB;
  constexpr int NA = BM / 8 / NW, NB = BN / 8 / NW;          // DMA instructions per wave per operand (8 rows x 128 B each)
  __shared__ __attribute__((aligned(1024))) char lds[NSTAGE * STAGE];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwg = gridDim.x, xcd = blockIdx.x & 7, xq = nwg >> 3, xr = nwg & 7;
  const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
  const int tm = wg / tiles_n, tn = wg % tiles_n;
  rsrc_t ra = make_rsrc(A, abytes), rb = make_rsrc(B, bbytes);
  uint32_t offa[NA], offb[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) offa[j] = (uint32_t)((tm * BM + (wave * NA + j) * 8 + lane / 8) * (size_t)pitch + (lane % 8) * 16);
#pragma unroll
  for (int j = 0; j < NB; ++j) offb[j] = (uint32_t)((tn * BN + (wave * NB + j) * 8 + lane / 8) * (size_t)pitch + (lane % 8) * 16);
  // register path: wave = (k-group kg, row half, column half wn); fragments j = 0, 1 (32 columns each), k-steps ks = 0, 1 of the k-group
  const int kg = wave >> 2, wn = wave & 1;
  const char* bp[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bp[j] = B + (size_t)(tn * BN + wn * 64 + 32 * j + (lane & 31)) * pitch + (kg * 2 * 16 + (lane >> 5) * 8) * 2;
  constexpr int LPT = MODE == 0 ? NA + NB : (MODE == 1 ? NA + 4 : (MODE == 2 ? NA : NB));          // loads per wave per tile
  u32x4 r[3][4];
  uint32_t acc = 0;
  int kcur = 0;
  auto issue = [&](int buf, u32x4 (&rr)[4]) {
    char* base = lds + buf * STAGE;
    const uint32_t kb = (uint32_t)kcur * RB;
    if (MODE != 3) {
#pragma unroll
      for (int j = 0; j < NA; ++j) dma16(ra, offa[j] + kb, base + (wave * NA + j) * 1024);
    }
    if (MODE == 0 || MODE == 3) {
#pragma unroll
      for (int j = 0; j < NB; ++j) dma16(rb, offb[j] + kb, base + BM * RB + (wave * NB + j) * 1024);
    } else if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) rr[j * 2 + ks] = gload(bp[j] + kb + ks * 32);
    }
    ++kcur;
  };
  issue(0, r[0]);
  issue(1, r[1]);
  // ktiles is a multiple of 3 here: the register sets rotate by name
  for (int t = 0; t < ktiles; t += 3) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int after = ktiles - 1 - (t + u);
      if (after >= 1) wait_vm<LPT>(); else wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      if (MODE == 1) { for (int q = 0; q < 4; ++q) use(r[u][q], acc); }
      if (t + u + 2 < ktiles) issue((u + 2) % 3, r[(u + 2) % 3]);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && (lds[123] == 77 || acc == 0x12345)) sink[0] = 1;
}

template <int MODE>
void run(char* A, char* B, int M, int N, int K, int* sink, int pad) {
  int tiles_m = (M + 127) / 128, tiles_n = (N + 127) / 128, tiles = tiles_m * tiles_n, ktiles = K * 2 / 128, pitch = K * 2 + pad;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i)
      hipLaunchKernelGGL((loop_kernel<MODE>), dim3(tiles), dim3(512), 0, 0, A, (uint32_t)((size_t)tiles_m * 128 * pitch), B, (uint32_t)((size_t)tiles_n * 128 * pitch), pitch, tiles_n, ktiles, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  printf("  mode %d pitch %5d: tiles %4d  k tiles %3d  %7.1f us  (%5.3f us per K tile)\n", MODE, pitch, tiles, ktiles, ms * 1e3, ms * 1e3 / ktiles);
}

int main() {
  char *A, *B; int* sink;
  hipMalloc(&A, 256u << 20); hipMalloc(&B, 64u << 20); hipMemset(A, 1, 256u << 20); hipMemset(B, 1, 64u << 20); hipMalloc(&sink, 4);
  int shapes[][3] = {{3840, 768, 3072}, {1920, 768, 3072}, {3840, 768, 768}, {3840, 768, 2304}};
  for (auto& s : shapes) {
    printf("M=%d N=%d K=%d (memory side of the 128 x 128 x 64 tile loop, 8 waves, 3 stages)\n", s[0], s[1], s[2]);
    for (int pad = 0; pad <= 128; pad += 128) {
      run<0>(A, B, s[0], s[1], s[2], sink, pad);
      run<1>(A, B, s[0], s[1], s[2], sink, pad);
      run<2>(A, B, s[0], s[1], s[2], sink, pad);
      run<3>(A, B, s[0], s[1], s[2], sink, pad);
    }
  }
  return 0;
}
