// Microbenchmark / layout probe for the block-scaled fp8 matrix instruction v_mfma_scale_f32_32x32x64_f8f6f4 (gfx950), which the guides give
// a rate for (twice the bf16 32x32x16 rate at e4m3 operands) but no operand map. Three questions, answered with exact integer data:
//   1. which k does byte j of lane l's 32-byte A / B operand hold? (hypotheses H1: k = 32 (l >> 5) + j; H2: two interleaved 16-byte halves,
//      k = 16 (l >> 5) + (j & 15) + 32 (j >> 4))
//   2. the scale operands: an E8M0 byte per lane (127 = 1.0), which byte op_sel picks, e5m2 through cbsz / blgp = 1
//   3. the issue rate against v_mfma_f32_32x32x16_bf16 and the non-scaled v_mfma_f32_32x32x16_fp8_fp8 (chip-wide, 4 waves per SIMD)
// Build: hipcc --offload-arch=gfx950 -O3 mfma_scale_probe.hip -o mfma_scale_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// e4m3 / e5m2 encodings of a few small values (exactly representable)
static uint8_t e4m3(float v) {
  // sign | 4-bit exponent (bias 7) | 3-bit mantissa; only exact small values are asked for
  if (v == 0.f) return 0;
  uint8_t s = v < 0 ? 0x80 : 0;
  float a = fabsf(v);
  int e;
  float m = frexpf(a, &e);          // a = m 2^e, m in [0.5, 1)
  m *= 2.f; e -= 1;                 // m in [1, 2)
  int mant = (int)((m - 1.f) * 8.f + 0.5f);
  return s | (uint8_t)((e + 7) << 3) | (uint8_t)mant;
}
static uint8_t e5m2(float v) {
  if (v == 0.f) return 0;
  uint8_t s = v < 0 ? 0x80 : 0;
  float a = fabsf(v);
  int e;
  float m = frexpf(a, &e);
  m *= 2.f; e -= 1;
  int mant = (int)((m - 1.f) * 4.f + 0.5f);
  return s | (uint8_t)((e + 15) << 2) | (uint8_t)mant;
}

template <int FA, int FB>
__global__ void one_mfma(const i32x8* a, const i32x8* b, f32x16* c, const int* sa, const int* sb, int sel) {
  f32x16 acc = {};
  const int l = threadIdx.x;
  if (sel == 0) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], acc, FA, FB, 0, sa[l], 0, sb[l]);
  else if (sel == 1) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], acc, FA, FB, 1, sa[l], 1, sb[l]);
  else if (sel == 2) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], acc, FA, FB, 2, sa[l], 2, sb[l]);
  else acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], acc, FA, FB, 3, sa[l], 3, sb[l]);
  c[l] = acc;
}

static int khyp(int hyp, int l, int j) {
  const int h = l >> 5;
  return hyp == 1 ? 32 * h + j : 16 * h + (j & 15) + 32 * (j >> 4);
}

template <int MODE>          // 0: scaled fp8 32x32x64, 1: bf16 32x32x16, 2: non-scaled fp8 32x32x16
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  i32x8 a8, b8;
  for (int r = 0; r < 8; ++r) { a8[r] = 0x38383838 + threadIdx.x; b8[r] = 0x30303030 + r; }
  bf16x8 ah, bh;
  for (int r = 0; r < 8; ++r) { ah[r] = (__bf16)(1.f + threadIdx.x * 0.01f); bh[r] = (__bf16)(0.5f + r); }
  long al = 0x3838383838383838l + threadIdx.x, bl = 0x3030303030303030l;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE == 0) acc[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[u], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      else if (MODE == 1) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[u], 0, 0, 0);
      else acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(al, bl, acc[u], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.f) out[0] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int FA, int FB>
static void layout_case(const char* name, int sel) {
  // A[i][k], B[k][j]: small exact values; the reference product in double
  static float A[32][64], B[64][32];
  srand(7 + FA * 2 + FB);
  const float vals[8] = {0.f, 1.f, -1.f, 2.f, 0.5f, -2.f, 1.5f, -0.5f};
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) A[i][k] = vals[rand() & 7];
  for (int k = 0; k < 64; ++k) for (int j = 0; j < 32; ++j) B[k][j] = vals[rand() & 7];
  for (int hyp = 1; hyp <= 2; ++hyp) {
    uint8_t ha[64][32], hb[64][32];
    int hsa[64], hsb[64];
    for (int l = 0; l < 64; ++l) {
      for (int j = 0; j < 32; ++j) {
        const int k = khyp(hyp, l, j);
        ha[l][j] = FA ? e5m2(A[l & 31][k]) : e4m3(A[l & 31][k]);
        hb[l][j] = FB ? e5m2(B[k][l & 31]) : e4m3(B[k][l & 31]);
      }
      // scale byte `sel` of the lane's scale register carries the lane's exponent: rows scaled by 2^(row & 1), columns by 2^-(col & 1)... per
      // LANE, i.e. per (row, 32-wide k block): the two k halves of a row get different scales, which the reference below applies
      const int ea = 127 + ((l & 1) ? 1 : 0) + ((l >> 5) ? 2 : 0), eb = 127 - ((l & 2) ? 1 : 0);
      hsa[l] = 0x01010101 * 99; hsb[l] = 0x01010101 * 99;          // the other bytes hold a wrong exponent on purpose
      hsa[l] = (hsa[l] & ~(0xFF << (8 * sel))) | (ea << (8 * sel));
      hsb[l] = (hsb[l] & ~(0xFF << (8 * sel))) | (eb << (8 * sel));
    }
    void *da, *db, *dc, *dsa, *dsb;
    CK(hipMalloc(&da, sizeof ha)); CK(hipMalloc(&db, sizeof hb)); CK(hipMalloc(&dc, 64 * 64)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256));
    CK(hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((one_mfma<FA, FB>), dim3(1), dim3(64), 0, 0, (const i32x8*)da, (const i32x8*)db, (f32x16*)dc, (const int*)dsa, (const int*)dsb, sel);
    CK(hipDeviceSynchronize());
    float hc[64][16];
    CK(hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost));
    int bad = 0, bad_noscale = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 16; ++r) {
        const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        double ref = 0, ref0 = 0;
        for (int k = 0; k < 64; ++k) {
          // the scale of A's lane (row, k block) and of B's lane (col, k block)
          const int la = row + 32 * (k >> 5), lb = col + 32 * (k >> 5);
          const double sa = ldexp(1.0, ((la & 1) ? 1 : 0) + ((la >> 5) ? 2 : 0)), sb = ldexp(1.0, -((lb & 2) ? 1 : 0));
          ref += (double)A[row][k] * B[k][col] * sa * sb;
          ref0 += (double)A[row][k] * B[k][col];
        }
        if (hc[l][r] != (float)ref) ++bad;
        if (hc[l][r] != (float)ref0) ++bad_noscale;
      }
    printf("%-28s op_sel %d  hypothesis H%d: %4d / 1024 wrong with per-lane scales, %4d wrong if scales were ignored\n", name, sel, hyp, bad, bad_noscale);
    hipFree(da); hipFree(db); hipFree(dc); hipFree(dsa); hipFree(dsb);
  }
}

template <int MODE>
static void rate_case(const char* name, double flop_per_mfma) {
  float* out;
  CK(hipMalloc(&out, 4));
  const int iters = 4096, grid = 256 * 4;           // 4 workgroups of 4 waves per CU: 4 waves per SIMD
  hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(256), 0, 0, out, 64);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double n = (double)grid * 4 * iters * 4;
  printf("%-44s %8.3f ms  %8.1f TFLOP/s  (%.1f ns per MFMA per SIMD at 4 waves)\n", name, best, n * flop_per_mfma / best / 1e9, best * 1e6 / ((double)iters * 4 * 4 * 4));
  hipFree(out);
}

int main() {
  for (int sel = 0; sel < 4; ++sel) layout_case<0, 0>("e4m3 x e4m3", sel);
  layout_case<1, 0>("e5m2 (A, cbsz 1) x e4m3", 0);
  layout_case<0, 1>("e4m3 x e5m2 (B, blgp 1)", 0);
  rate_case<1>("v_mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16);
  rate_case<2>("v_mfma_f32_32x32x16_fp8_fp8", 2.0 * 32 * 32 * 16);
  rate_case<0>("v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3)", 2.0 * 32 * 32 * 64);
  return 0;
}
