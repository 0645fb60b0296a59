// Microbenchmark: HBM streaming rate of the BatchNorm-backward dgrad epilogue's access pattern by ROW-SEGMENT width. 512 persistent workgroups
// (two per CU) walk tiles of R rows x S bytes of three [M][C] bf16 tensors (read y, read res, write out = y + res), eight rows per thread and
// tile, four rows of requests in flight — what igemm_epilogue_bn does after its main loop — with S = 256 B (the 128-column tile), 512 B, 1 KB
// and the whole row. Build on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/micro/seg_rate.hip -o /tmp/seg_rate && /tmp/seg_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rsrc_t make_rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)0xFFFFFFFF, 0x00020000); }
__device__ __forceinline__ u32x4 ld16(rsrc_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); }
__device__ __forceinline__ void st16(rsrc_t r, uint32_t off, u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0); }

// SEG: bytes of one row segment per tile; a tile is 32768 / SEG rows (32 KB per tensor and tile, as the 128 x 128 bf16 tile)
template <int SEG, int AHEAD>
__global__ __launch_bounds__(256, 2) void seg_kernel(const void* y, const void* res, void* out, int M, int pitch, int tiles_per_wg) {
  constexpr int LPR = SEG / 16, RPS = 256 / LPR, ROWS = 32768 / SEG, RPT = ROWS / RPS;      // lanes per row, rows per sweep, rows per tile, rows per thread
  const rsrc_t ry = make_rsrc(y), rr = make_rsrc(res), ro = make_rsrc(out);
  const int segs = pitch / SEG;                      // column tiles
  const int tid = threadIdx.x;
  const int ecol = (tid % LPR) * 16, erow0 = tid / LPR;
  // workgroup -> (row slice, column tile): column tiles of one row range adjacent, like igemm_dma_bn_kernel
  const int wg = blockIdx.x, slice = wg / segs, ct = wg - slice * segs;
  const int row_begin = slice * tiles_per_wg * ROWS;
  for (int t = 0; t < tiles_per_wg; ++t) {
    const int m0 = row_begin + t * ROWS;
    if (m0 >= M) break;
    u32x4 a[RPT], b[RPT];
    uint32_t off[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const int row = m0 + erow0 + q * RPS;
      off[q] = row < M ? (uint32_t)row * (uint32_t)pitch + (uint32_t)(ct * SEG + ecol) : 0x80000000u;
    }
#pragma unroll
    for (int q = 0; q < AHEAD && q < RPT; ++q) { a[q] = ld16(ry, off[q]); b[q] = ld16(rr, off[q]); }
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      if (q + AHEAD < RPT) { a[q + AHEAD] = ld16(ry, off[q + AHEAD]); b[q + AHEAD] = ld16(rr, off[q + AHEAD]); }
      u32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = a[q][e] + (b[q][e] & 0x00010001u);
      st16(ro, off[q], v);
    }
  }
}

template <int SEG, int AHEAD>
static void run(const void* y, const void* res, void* out, int M, int C, const char* label) {
  const int pitch = C * 2, segs = pitch / SEG, rows_per_tile = 32768 / SEG;
  const int slices = 512 / segs;
  const int tiles = (M + rows_per_tile - 1) / rows_per_tile;
  const int tpw = (tiles + slices - 1) / slices;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int it = 0; it < 5; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((seg_kernel<SEG, AHEAD>), dim3(slices * segs), dim3(256), 0, 0, y, res, out, M, pitch, tpw);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (it && ms < best) best = ms;
  }
  const double bytes = 3.0 * M * pitch;
  printf("M=%7d C=%5d  %-28s %7.1f us  %6.0f GB/s\n", M, C, label, best * 1e3, bytes / best / 1e6);
}

int main() {
  const size_t cap = (size_t)401408 * 256 * 2;
  void *y, *res, *out;
  hipMalloc(&y, cap); hipMalloc(&res, cap); hipMalloc(&out, cap);
  hipMemset(y, 1, cap); hipMemset(res, 2, cap); hipMemset(out, 0, cap);
  struct { int M, C; } shapes[] = {{401408, 256}, {100352, 512}, {25088, 1024}, {401408, 64}};
  for (auto s : shapes) {
    if (s.C * 2 >= 256) run<256, 4>(y, res, out, s.M, s.C, "256 B segments, 4 ahead");
    if (s.C * 2 >= 256) run<256, 8>(y, res, out, s.M, s.C, "256 B segments, 8 ahead");
    if (s.C * 2 >= 512) run<512, 4>(y, res, out, s.M, s.C, "512 B segments, 4 ahead");
    if (s.C * 2 >= 1024) run<1024, 4>(y, res, out, s.M, s.C, "1 KB segments, 4 ahead");
    if (s.C * 2 >= 2048) run<2048, 4>(y, res, out, s.M, s.C, "2 KB segments, 4 ahead");
    if (s.C * 2 == 128) run<128, 4>(y, res, out, s.M, s.C, "128 B segments (whole rows)");
  }
  return 0;
}
