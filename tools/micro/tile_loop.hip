// Microbenchmark: memory side of the implicit-GEMM main loop for candidate tile shapes. One workgroup of NW waves per BM x BN output tile
// streams K tiles of A (BM rows x RB bytes) and B (BN rows x RB bytes) through an NSTAGE LDS ring with `buffer_load ... lds`, counted
// vmcnt and one s_barrier per K tile (no LDS reads, no MFMAs). Answers: what does a 256-row tile / a 128-byte K slab buy on the step's shapes?
// Build on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/micro/tile_loop.hip -o tools/micro/tile_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ void dma16(rsrc_t r, uint32_t off, void* lds) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BM, int BN, int RB, int NW, int NSTAGE>
__global__ __launch_bounds__(NW * 64) void loop_kernel(const char* A, uint32_t abytes, const char* B, uint32_t bbytes, int pitch, int tiles_n, int ktiles, int* sink) {
  constexpr int STAGE = (BM + BN) * RB;
  constexpr int RPI = 1024 / RB;                 // rows per DMA instruction
  constexpr int CPR = RB / 16;                   // 16-B chunks per row
  constexpr int NA = BM / RPI / NW, NB = BN / RPI / NW;
  static_assert(NA >= 1 && NB >= 1, "tile too small for the wave count");
  __shared__ __attribute__((aligned(1024))) char lds[NSTAGE * STAGE];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwg = gridDim.x, xcd = blockIdx.x & 7, xq = nwg >> 3, xr = nwg & 7;
  const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
  const int tm = wg / tiles_n, tn = wg % tiles_n;
  rsrc_t ra = make_rsrc(A, abytes), rb = make_rsrc(B, bbytes);
  uint32_t offa[NA], offb[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) offa[j] = (uint32_t)((tm * BM + (wave * NA + j) * RPI + lane / CPR) * (size_t)pitch + (lane % CPR) * 16);
#pragma unroll
  for (int j = 0; j < NB; ++j) offb[j] = (uint32_t)((tn * BN + (wave * NB + j) * RPI + lane / CPR) * (size_t)pitch + (lane % CPR) * 16);
  int kcur = 0;
  auto issue = [&](int buf) {
    char* base = lds + buf * STAGE;
    const uint32_t kb = (uint32_t)kcur * RB;
#pragma unroll
    for (int j = 0; j < NA; ++j) dma16(ra, offa[j] + kb, base + (wave * NA + j) * 1024);
#pragma unroll
    for (int j = 0; j < NB; ++j) dma16(rb, offb[j] + kb, base + BM * RB + (wave * NB + j) * 1024);
    ++kcur;
  };
#pragma unroll
  for (int p = 0; p < NSTAGE - 1; ++p) if (p < ktiles) issue(p);
  int buf = 0;
  for (int t = 0; t < ktiles; ++t) {
    const int after = ktiles - 1 - t;
    if (NSTAGE >= 4 && after >= 2) wait_vm<2 * (NA + NB)>();
    else if (NSTAGE >= 3 && after >= 1) wait_vm<NA + NB>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (t + NSTAGE - 1 < ktiles) { int nb = buf + NSTAGE - 1; if (nb >= NSTAGE) nb -= NSTAGE; issue(nb); }
    if (++buf == NSTAGE) buf = 0;
  }
  __syncthreads();
  if (threadIdx.x == 0 && lds[123] == 77) sink[0] = 1;
}

template <int BM, int BN, int RB, int NW, int NSTAGE>
void run(char* A, char* B, int M, int N, int K, int* sink) {
  int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, tiles = tiles_m * tiles_n, ktiles = K * 2 / RB, pitch = K * 2;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i)
      hipLaunchKernelGGL((loop_kernel<BM, BN, RB, NW, NSTAGE>), dim3(tiles), dim3(NW * 64), 0, 0, A, (uint32_t)((size_t)tiles_m * BM * pitch), B, (uint32_t)((size_t)tiles_n * BN * pitch), pitch, tiles_n, ktiles, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  double gb = (double)tiles * ktiles * (BM + BN) * RB / 1e9;
  double tf = 2.0 * M * N * (double)K / (ms * 1e-3) / 1e12;
  printf("  %3dx%3d rowB %3d waves %d stages %d: tiles %5d  %7.1f us  %6.1f TB/s staged  (= %6.0f TF/s if the MFMAs hid behind it)\n", BM, BN, RB, NW, NSTAGE, tiles, ms * 1e3, gb / ms, tf);
}

int main() {
  char *A, *B; int* sink;
  hipMalloc(&A, 1024u << 20); hipMalloc(&B, 256u << 20); hipMemset(A, 1, 1024u << 20); hipMemset(B, 1, 256u << 20); hipMalloc(&sink, 4);
  // M, N, K of representative launches: BERT FFN2 / FFN1 / QKV, layer3 3x3 conv, layer2 3x3, layer3 1x1 expand, layer4 1x1 reduce, layer2 1x1 expand
  int shapes[][3] = {{3840, 768, 3072}, {3840, 3072, 768}, {3840, 2304, 768}, {25088, 256, 2304}, {100352, 128, 1152}, {25088, 1024, 256}, {6272, 512, 2048}, {100352, 512, 128}};
  for (auto& s : shapes) {
    printf("M=%d N=%d K=%d\n", s[0], s[1], s[2]);
    run<128, 128, 64, 4, 3>(A, B, s[0], s[1], s[2], sink);
    run<128, 128, 64, 4, 4>(A, B, s[0], s[1], s[2], sink);
    run<128, 128, 128, 4, 2>(A, B, s[0], s[1], s[2], sink);
    run<128, 128, 128, 4, 3>(A, B, s[0], s[1], s[2], sink);
    run<256, 128, 64, 8, 3>(A, B, s[0], s[1], s[2], sink);
    run<256, 128, 64, 4, 3>(A, B, s[0], s[1], s[2], sink);
    run<256, 128, 128, 8, 2>(A, B, s[0], s[1], s[2], sink);
    run<256, 128, 128, 8, 3>(A, B, s[0], s[1], s[2], sink);
    run<256, 256, 64, 8, 3>(A, B, s[0], s[1], s[2], sink);
    run<256, 256, 128, 8, 2>(A, B, s[0], s[1], s[2], sink);
  }
  return 0;
}
