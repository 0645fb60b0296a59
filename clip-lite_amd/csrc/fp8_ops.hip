// OCP e4m3 forward path (BASELINE.json configs[4]: "fp8 weights/activations on CDNA4 fp8 MFMA"; the reference has no fp8 semantics —
// the policy is this build's, stated in DESIGN.md): per-tensor current scaling, quantised operands for the forward conv / linear GEMMs,
// f32 accumulation on v_mfma_f32_32x32x16_fp8_fp8, bf16 outputs; backward stays bf16 on the saved bf16 activations.
//
//   clite_fp8_quantize : amax = max|x| (integer atomic max on the f32 bit pattern: order-independent), scale = 448 / amax,
//                        q = e4m3(clamp(x * scale, +-448)); scales = {scale, 1 / scale}
//   clite_gemm_nt_fp8 / clite_conv_fwd_fp8 : igemm_dma_kernel's pipeline over 64-byte K slabs = 64 fp8 elements (K tile 64), fragments by
//                        ds_read_b64 (8 fp8 per lane and k-step), epilogue of igemm.h applied to acc * a_scales[1] * b_scales[1].
#include "igemm_dma.h"
#include "clite.h"
#include "wide_api.h"

using namespace clite;

namespace {

typedef uint8_t fp8;
constexpr float FP8_MAX = 448.f;

template <typename T>
__global__ __launch_bounds__(256) void amax_kernel(const T* __restrict__ x, size_t n8, float* amax) {
  float m = 0.f;
  bool nan = false;       // fmaxf drops a NaN operand: a diverged tensor must not come out as a finite amax (it is carried separately)
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    float v[8];
    load8(x + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) { m = fmaxf(m, fabsf(v[e])); nan = nan || v[e] != v[e]; }
  }
  __shared__ uint32_t red[4];
  // the quiet-NaN pattern 0x7FC00000 orders above every finite value and above +inf both as a float bit pattern under the integer max
  // below and in this wave / workgroup fold (done on the bit patterns for that reason)
  uint32_t mb = nan ? 0x7FC00000u : f32_bits(m);
#pragma unroll
  for (int sh = 32; sh >= 1; sh >>= 1) { const uint32_t o = (uint32_t)wave_shfl_xor_i((int)mb, sh); mb = o > mb ? o : mb; }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mb;
  __syncthreads();
  // non-negative floats order like their bit patterns: an integer max needs no float atomics and is order-independent. One atomic per
  // workgroup: same-address atomics retire at ~90 per microsecond chip-wide, so thousands of them per call would cost more than the pass
  if (threadIdx.x == 0) {
    uint32_t b = red[0];
    for (int w = 1; w < 4; ++w) b = red[w] > b ? red[w] : b;
    atomic_max_u32((uint32_t*)amax, b);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void to_fp8_kernel(const T* __restrict__ x, size_t n8, const float* __restrict__ amax, fp8* __restrict__ out, float* scales) {
  const float a = amax[0];
  // a non-finite amax (a NaN or an infinity somewhere in the tensor) makes both scales NaN, so every product of the GEMM that consumes them
  // is NaN: a diverged operand stays visible in the loss instead of being clamped into a finite e4m3 value
  const bool finite = a == a && a < INFINITY;
  const float scale = !finite ? bits_f32(0x7FC00000u) : (a > 0.f ? FP8_MAX / a : 1.f);
  if (blockIdx.x == 0 && threadIdx.x == 0) { scales[0] = scale; scales[1] = !finite ? scale : (a > 0.f ? a / FP8_MAX : 1.f); }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    float v[8];
    load8(x + i * 8, v);
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      // fminf / fmaxf would turn a NaN into -448: a NaN passes the clamp untouched and converts to e4m3's NaN (0x7F / 0xFF)
      const float p0 = v[2 * e] * scale, q0 = v[2 * e + 1] * scale;
      const float p = p0 != p0 ? p0 : fminf(fmaxf(p0, -FP8_MAX), FP8_MAX), q = q0 != q0 ? q0 : fminf(fmaxf(q0, -FP8_MAX), FP8_MAX);
      w[e] = cvt2_fp8(p, q);
    }
    *(u32x2*)(out + i * 8) = u32x2{w[0] | (w[1] << 16), w[2] | (w[3] << 16)};
  }
}

// ---- delayed scaling (the producer-fused quantiser: bn_apply writes the e4m3 copy with LAST step's scale and records THIS step's amax)
__global__ void scale_update_kernel(float* amax, float* scales, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t* slot = (uint32_t*)amax + (size_t)i * (CLITE_FP8_AMAX_REPLICAS * CLITE_FP8_AMAX_STRIDE);
  uint32_t ab = 0;
  for (int r = 0; r < CLITE_FP8_AMAX_REPLICAS; ++r) { const uint32_t v = slot[r * CLITE_FP8_AMAX_STRIDE]; ab = v > ab ? v : ab; slot[r * CLITE_FP8_AMAX_STRIDE] = 0u; }
  const float a = bits_f32(ab);
  if (ab == 0u) return;          // nothing recorded this step: keep the scales
  const bool finite = a == a && a < INFINITY;
  const float nanv = bits_f32(0x7FC00000u);
  scales[2 * i] = finite ? FP8_MAX / a : nanv;
  scales[2 * i + 1] = finite ? a / FP8_MAX : nanv;
}

// ---- many tensors of one arena, per-tensor current scaling, two launches (the conv weights of a step)
constexpr int GROUP_CHUNK = 8192;          // elements per workgroup: 256 threads x 8 elements x 4 sweeps
DEV void group_span(const clite_fp8_item* items, const uint32_t* table, int& item, size_t& lo8, size_t& hi8) {
  const uint32_t e = table[blockIdx.x];
  item = (int)(e >> 12);
  const size_t begin = (size_t)(e & 0xFFFu) * GROUP_CHUNK, numel = items[item].numel;
  const size_t end = begin + GROUP_CHUNK < numel ? begin + GROUP_CHUNK : numel;
  lo8 = (items[item].offset + begin) / 8;
  hi8 = (items[item].offset + end) / 8;
}
// pass 1: every workgroup leaves the max |x| of its chunk in partial[blockIdx.x] (bit pattern; NaN -> the quiet-NaN pattern) — no atomics and
// nothing to zero, so the two launches are a pure function of the arena (same bits every time, and no memset node inside a captured step)
__global__ __launch_bounds__(256) void group_amax_kernel(const bf16* __restrict__ base, const clite_fp8_item* items, const uint32_t* table, uint32_t* partial) {
  int item; size_t lo8, hi8;
  group_span(items, table, item, lo8, hi8);
  float m = 0.f;
  bool nan = false;
  for (size_t i = lo8 + threadIdx.x; i < hi8; i += 256) {
    float v[8];
    load8(base + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) { m = fmaxf(m, fabsf(v[e])); nan = nan || v[e] != v[e]; }
  }
  __shared__ uint32_t red[4];
  uint32_t mb = nan ? 0x7FC00000u : f32_bits(m);          // (amax_kernel: bit patterns order like the values, the NaN pattern above all)
#pragma unroll
  for (int sh = 32; sh >= 1; sh >>= 1) { const uint32_t o = (uint32_t)wave_shfl_xor_i((int)mb, sh); mb = o > mb ? o : mb; }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mb;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t b = red[0];
    for (int w = 1; w < 4; ++w) b = red[w] > b ? red[w] : b;
    partial[blockIdx.x] = b;
  }
}
// pass 2: the item's amax = max over its chunks' partials (the table lists an item's chunks consecutively: its first workgroup is
// blockIdx.x - chunk), then the conversion; the workgroup of chunk 0 publishes amax and scales
__global__ __launch_bounds__(256) void group_to_fp8_kernel(const bf16* __restrict__ base, const clite_fp8_item* items, const uint32_t* table, const uint32_t* __restrict__ partial,
                                                           fp8* __restrict__ q, float* amax, float* scales) {
  int item; size_t lo8, hi8;
  group_span(items, table, item, lo8, hi8);
  const int chunk = (int)(table[blockIdx.x] & 0xFFFu), nchunks = (int)((items[item].numel + GROUP_CHUNK - 1) / GROUP_CHUNK);
  const uint32_t* mine = partial + (blockIdx.x - chunk);
  uint32_t ab = 0;
  for (int c = threadIdx.x & 63; c < nchunks; c += 64) ab = mine[c] > ab ? mine[c] : ab;
#pragma unroll
  for (int sh = 32; sh >= 1; sh >>= 1) { const uint32_t o = (uint32_t)wave_shfl_xor_i((int)ab, sh); ab = o > ab ? o : ab; }
  const float a = bits_f32(ab);
  if (chunk == 0 && threadIdx.x == 0) amax[item] = a;
  const bool finite = a == a && a < INFINITY;
  const float scale = !finite ? bits_f32(0x7FC00000u) : (a > 0.f ? FP8_MAX / a : 1.f);
  if (chunk == 0 && threadIdx.x == 0) { scales[2 * item] = scale; scales[2 * item + 1] = !finite ? scale : (a > 0.f ? a / FP8_MAX : 1.f); }
  for (size_t i = lo8 + threadIdx.x; i < hi8; i += 256) {
    float v[8];
    load8(base + i * 8, v);
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float p0 = v[2 * e] * scale, q0 = v[2 * e + 1] * scale;
      const float pp = p0 != p0 ? p0 : fminf(fmaxf(p0, -FP8_MAX), FP8_MAX), qq = q0 != q0 ? q0 : fminf(fmaxf(q0, -FP8_MAX), FP8_MAX);
      w[e] = cvt2_fp8(pp, qq);
    }
    *(u32x2*)(q + i * 8) = u32x2{w[0] | (w[1] << 16), w[2] | (w[3] << 16)};
  }
}

// ---- GEMM on fp8 operands. KC x KC only (both operands k-contiguous: activations x weights, the forward direction).
// LDS image: igemm_dma.h's KC image [rows][64 B] with 16-byte chunk c of row r in slot c ^ ((r>>2)&3); a k-step is 16 fp8 = one chunk, and
// lane (r, h) reads bytes 8h..8h+7 of it.
template <int ROWS> using FKC = DmaKC<fp8, ROWS, 64, false>;
DEV int fp8_frag_off(int x0, int ks, int lane) {
  const int r = x0 + (lane & 31);
  return r * 64 + ((ks ^ ((r >> 2) & 3)) << 4) + 8 * (lane >> 5);
}

// (fp8_frag_off64, the image read of the block-scaled instruction: igemm_dma.h)
// SCALED: the block-scaled instruction at unit scales (twice the matrix rate; default) / the non-scaled 32x32x16 form (-DCLITE_FP8_SCALED=0, A/B)
template <class CFG, int ROWS_A, int ROWS_B, int EPI, bool SCALED>
__global__ __launch_bounds__(256) void igemm_fp8_kernel(FKC<ROWS_A> la, FKC<ROWS_B> lb, Epilogue ep, RowMap rm, const float* sa_scales, const float* sb_scales,
                                                        int M, int N, int ktiles) {
  typedef FKC<ROWS_A> LA;
  typedef FKC<ROWS_B> LB;
  constexpr int BM = CFG::BM, BN = CFG::BN;
  constexpr int RM = CFG::RM, RN = CFG::RN;
  constexpr int NSTAGE = 3, KSTEPS = 4;
  constexpr int STAGE = LA::BYTES + LB::BYTES;
  constexpr int SMEM = (NSTAGE * STAGE > CFG::EPI_BYTES) ? NSTAGE * STAGE : CFG::EPI_BYTES;
  constexpr int LOADS_PER_TILE = LA::NI + LB::NI;
  __shared__ __attribute__((aligned(1024))) char smem[SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = wave_uniform(tid >> 6);
  const int wm0 = (wave / CFG::WAVES_N) * CFG::WM;
  const int wn0 = (wave % CFG::WAVES_N) * CFG::WN;
  const int tiles_n = (N + BN - 1) / BN;
  const int nwg = gridDim.x, xcd = blockIdx.x & 7, xq = nwg >> 3, xr = nwg & 7;
  const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  ep.alpha *= sa_scales[1] * sb_scales[1];                 // de-quantisation: 1/scale of both operands

  typename LA::State sa;
  typename LB::State sb;
  la.init(sa, m0, wave, lane, 0);
  lb.init(sb, n0, wave, lane, 0);
  f32x16 acc[RM][RN];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int aoff[RM][KSTEPS], boff[RN][KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
    for (int i = 0; i < RM; ++i) aoff[i][ks] = SCALED ? fp8_frag_off64(wm0 + i * 32, ks & 1, lane) : fp8_frag_off(wm0 + i * 32, ks, lane);
#pragma unroll
    for (int j = 0; j < RN; ++j) boff[j][ks] = SCALED ? fp8_frag_off64(wn0 + j * 32, ks & 1, lane) : fp8_frag_off(wn0 + j * 32, ks, lane);
  }
#pragma unroll
  for (int pz = 0; pz < NSTAGE - 1; ++pz) {
    if (pz < ktiles) {
      la.issue(sa, smem + pz * STAGE, wave);
      lb.issue(sb, smem + pz * STAGE + LA::BYTES, wave);
    }
  }
  int buf = 0;
  for (int t = 0; t < ktiles; ++t) {
    if (ktiles - 1 - t >= 1) wait_vmcnt<LOADS_PER_TILE>();
    else wait_vmcnt<0>();
    barrier_raw();
    const char* abuf = smem + buf * STAGE;
    const char* bbuf = abuf + LA::BYTES;
    if constexpr (SCALED) {
      u32x4 alo[RM], ahi[RM], blo[RN], bhi[RN];
#pragma unroll
      for (int i = 0; i < RM; ++i) { alo[i] = *(const u32x4*)(abuf + aoff[i][0]); ahi[i] = *(const u32x4*)(abuf + aoff[i][1]); }
#pragma unroll
      for (int j = 0; j < RN; ++j) { blo[j] = *(const u32x4*)(bbuf + boff[j][0]); bhi[j] = *(const u32x4*)(bbuf + boff[j][1]); }
      if (t + NSTAGE - 1 < ktiles) {
        int nb = buf + NSTAGE - 1; if (nb >= NSTAGE) nb -= NSTAGE;
        la.issue(sa, smem + nb * STAGE, wave);
        lb.issue(sb, smem + nb * STAGE + LA::BYTES, wave);
      }
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) acc[i][j] = mfma32x64_fp8(alo[i], ahi[i], blo[j], bhi[j], acc[i][j]);
      if (++buf == NSTAGE) buf = 0;
      continue;
    }
    uint64_t af0[RM], bf0[RN];
#pragma unroll
    for (int i = 0; i < RM; ++i) af0[i] = *(const uint64_t*)(abuf + aoff[i][0]);
#pragma unroll
    for (int j = 0; j < RN; ++j) bf0[j] = *(const uint64_t*)(bbuf + boff[j][0]);
    if (t + NSTAGE - 1 < ktiles) {
      int nb = buf + NSTAGE - 1; if (nb >= NSTAGE) nb -= NSTAGE;
      la.issue(sa, smem + nb * STAGE, wave);
      lb.issue(sb, smem + nb * STAGE + LA::BYTES, wave);
    }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      uint64_t af[RM], bfr[RN];
#pragma unroll
      for (int i = 0; i < RM; ++i) af[i] = ks == 0 ? af0[i] : *(const uint64_t*)(abuf + aoff[i][ks]);
#pragma unroll
      for (int j = 0; j < RN; ++j) bfr[j] = ks == 0 ? bf0[j] : *(const uint64_t*)(bbuf + boff[j][ks]);
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) acc[i][j] = mfma32_fp8(af[i], bfr[j], acc[i][j]);
    }
    if (++buf == NSTAGE) buf = 0;
  }
  barrier_raw();
  if constexpr (EPI == 2) igemm_epilogue_plain<bf16, CFG>(acc, ep, rm, smem, M, N, m0, n0, tid, lane, wave, wm0, wn0);
  else igemm_epilogue<bf16, CFG, false, true>(acc, ep, rm, smem, M, N, m0, n0, tid, lane, wave, wm0, wn0);
}

FastDiv fastdiv_make(uint32_t d) {
  FastDiv f;
  f.d = d;
  if (d <= 1) { f.mul = 0; f.shift = 0; f.d = 1; return f; }
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;
  f.shift = l;
  f.mul = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
  return f;
}
ConvGeom geom_dense(int rows, int K, int ld) {
  ConvGeom g;
  g.H = 1; g.W = 1; g.C = K;
  g.sN = ld; g.sH = 0; g.sW = 0;
  g.RH = 1; g.RW = 1; g.R = 1; g.S = 1; g.stride = 1; g.pad = 0; g.padw = 0;
  g.rows = rows; g.concat = 0;
  g.div_hw = fastdiv_make(1);
  g.div_w = fastdiv_make(1);
  return g;
}
ConvGeom geom_fwd(const clite_conv& c) {
  ConvGeom g;
  g.H = c.H; g.W = c.W; g.C = c.C;
  g.sN = c.H * c.W * c.C; g.sH = c.W * c.C; g.sW = c.C;
  g.RH = c.Ho; g.RW = c.Wo; g.R = c.R; g.S = c.S; g.stride = c.stride; g.pad = c.pad; g.padw = c.pad;
  g.rows = c.N * c.Ho * c.Wo; g.concat = 0;
  g.div_hw = fastdiv_make(c.Ho * c.Wo);
  g.div_w = fastdiv_make(c.Wo);
  return g;
}

typedef TileCfg<128, 128, 64, 64, 64> F128;       // BK = 64 fp8 elements = 64 bytes per row
typedef TileCfg<256, 64, 64, 64, 64> F256x64;
typedef TileCfg<256, 128, 64, 128, 64> F256x128;

bool plain(const clite_epilogue& ep) {
  return !ep.out_f32 && !ep.preact && ep.act == CLITE_ACT_NONE && !ep.dact_aux && ep.drop_p <= 0.f && !ep.residual && !ep.fp8_out && !ep.fp8_amax;
}
int check_ep8(const clite_epilogue* ep, int N, bool q8 = false) {
  if (!ep || !ep->out || ep->atomic || ep->bn_y || ep->mask_after_residual || ep->splitk_ws || ep->residual_subsample > 1) return -1;
  if (N % 8 || ep->ldc % 8) return -1;
  if ((ep->fp8_out || ep->fp8_amax || ep->fp8_scale) && (!q8 || ep->out_f32 || (ep->fp8_out && !ep->fp8_scale))) return -1;
  return 0;
}

#ifndef CLITE_FP8_SCALED
#define CLITE_FP8_SCALED 1
#endif
template <class CFG, int RA, int RB>
void go8(const void* A, uint32_t ab, const ConvGeom& ga, const void* B, uint32_t bb, const ConvGeom& gb, const clite_epilogue& ep, const float* sa,
         const float* sb, int M, int N, int ktiles, hipStream_t st) {
  RowMap rm{};
  const int tiles = ((M + CFG::BM - 1) / CFG::BM) * ((N + CFG::BN - 1) / CFG::BN);
  FKC<RA> la{A, ab, ga};
  FKC<RB> lb{B, bb, gb};
  constexpr bool S = CLITE_FP8_SCALED != 0;
  if (plain(ep)) hipLaunchKernelGGL((igemm_fp8_kernel<CFG, RA, RB, 2, S>), dim3(tiles), dim3(256), 0, st, la, lb, ep, rm, sa, sb, M, N, ktiles);
  else hipLaunchKernelGGL((igemm_fp8_kernel<CFG, RA, RB, 0, S>), dim3(tiles), dim3(256), 0, st, la, lb, ep, rm, sa, sb, M, N, ktiles);
}

int launch8(const void* A, uint32_t ab, const ConvGeom& ga, const void* B, uint32_t bb, const ConvGeom& gb, const clite_epilogue& ep, const float* sa,
            const float* sb, int M, int N, int Ktot, hipStream_t st) {
  const int ktiles = (Ktot + 63) / 64;
  // tile: 256 x 64 for narrow outputs; otherwise 128 x 128, or 256 x 128 (one instruction of the scaled form does the work of four, so a K tile of
  // the 128 x 128 tile is 4 instructions per wave between two barriers: the taller tile halves the barriers and the L2 -> LDS bytes per product)
  // for deep-K launches that still fill the chip with it (tools/probe_fp8.py, batch 256: 3 x 3 256 -> 256 at 14 x 14 48.1 -> 38.9 us, 1 x 1
  // 1024 -> 256 27.8 -> 25.5; batch 128: 3 x 3 128 -> 128 at 28 x 28 32.1 -> 27.5; launches of < 256 tall tiles lose 15 - 45 %).
  // clite_set_tile_policy: 1 forces 128 x 128, 2 forces 256 x 128.
  const int pol = tile_policy_value();
  const long tiles_tall = (long)((M + 255) / 256) * ((N + 127) / 128);
  if (N <= 64) go8<F256x64, 256, 64>(A, ab, ga, B, bb, gb, ep, sa, sb, M, N, ktiles, st);
  else if (pol == 2 || (pol == 0 && CLITE_FP8_SCALED && tiles_tall >= 320 && Ktot >= 1024)) go8<F256x128, 256, 128>(A, ab, ga, B, bb, gb, ep, sa, sb, M, N, ktiles, st);
  else go8<F128, 128, 128>(A, ab, ga, B, bb, gb, ep, sa, sb, M, N, ktiles, st);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int clite_fp8_quantize(int dtype, const void* x, uint64_t n, float* amax, float* scales, void* out_fp8, void* stream) {
  if (!x || !amax || !scales || !out_fp8 || n == 0 || n % 8) return -1;
  hipStream_t st = (hipStream_t)stream;
  const size_t n8 = n / 8;
  size_t g = (n8 + 255) / 256;
  const int grid = (int)(g < 1024 ? g : 1024);
  if (dtype == CLITE_BF16) {
    hipLaunchKernelGGL(amax_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)x, n8, amax);
    hipLaunchKernelGGL(to_fp8_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)x, n8, (const float*)amax, (fp8*)out_fp8, scales);
  } else if (dtype == CLITE_F32) {
    hipLaunchKernelGGL(amax_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, n8, amax);
    hipLaunchKernelGGL(to_fp8_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, n8, (const float*)amax, (fp8*)out_fp8, scales);
  } else {
    return -1;
  }
  return (int)hipGetLastError();
}

extern "C" int clite_fp8_scale_update(float* amax, float* scales, int n, void* stream) {
  if (!amax || !scales || n <= 0) return -1;
  hipLaunchKernelGGL(scale_update_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, amax, scales, n);
  return (int)hipGetLastError();
}

extern "C" int clite_fp8_quantize_group(const void* base, const clite_fp8_item* items_dev, const uint32_t* wg_table_dev, int n_items, int n_wgs,
                                        uint32_t* partial, float* amax, float* scales, void* q_base, void* stream) {
  if (!base || !items_dev || !wg_table_dev || !partial || !amax || !scales || !q_base || n_items <= 0 || n_items >= (1 << 20) || n_wgs <= 0) return -1;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(group_amax_kernel, dim3(n_wgs), dim3(256), 0, st, (const bf16*)base, items_dev, wg_table_dev, partial);
  hipLaunchKernelGGL(group_to_fp8_kernel, dim3(n_wgs), dim3(256), 0, st, (const bf16*)base, items_dev, wg_table_dev, (const uint32_t*)partial, (fp8*)q_base, amax, scales);
  return (int)hipGetLastError();
}

extern "C" int clite_gemm_nt_fp8(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const float* a_scales, const float* b_scales,
                                 const clite_epilogue* ep, void* stream) {
  if (!A || !B || !a_scales || !b_scales || M <= 0 || N <= 0 || K <= 0 || K % 16 || lda % 16 || ldb % 16 || lda < K || ldb < K || check_ep8(ep, N, true)) return -1;
  if ((size_t)M * lda >= 0xF0000000ull || (size_t)N * ldb >= 0xF0000000ull) return -1;
  const uint32_t ab = (uint32_t)((size_t)(M - 1) * lda + K), bb = (uint32_t)((size_t)(N - 1) * ldb + K);
  return launch8(A, ab, geom_dense(M, K, lda), B, bb, geom_dense(N, K, ldb), *ep, a_scales, b_scales, M, N, K, (hipStream_t)stream);
}

// Input gradient of a stride-1 convolution on fp8 operands, in the BatchNorm-backward form of the ResNet backward (igemm_dma_bn_kernel's FORM 1: packed
// relu' bits, the BatchNorm input, bf16 output, both reductions): A = dy in e5m2 gathered with the transposed-conv geometry, B = the transposed weights
// [C][R][S][K] in e4m3. One 128 x 128 tile per workgroup.
ConvGeom geom_dgrad8(const clite_conv& c) {
  ConvGeom g;
  g.H = c.Ho; g.W = c.Wo; g.C = c.K;
  g.sN = c.Ho * c.Wo * c.K; g.sH = c.Wo * c.K; g.sW = c.K;
  g.RH = c.H; g.RW = c.W; g.R = c.R; g.S = c.S; g.stride = c.stride; g.pad = c.pad; g.padw = c.pad;
  g.rows = c.N * c.H * c.W; g.concat = 0;
  g.div_hw = fastdiv_make(c.H * c.W);
  g.div_w = fastdiv_make(c.W);
  return g;
}

extern "C" int clite_conv_dgrad_fp8(const void* dy8, const void* wt8, const clite_conv* cv, const float* dy_scales, const float* w_scales,
                                    const clite_epilogue* ep, void* stream) {
  if (!dy8 || !wt8 || !cv || !dy_scales || !w_scales || !ep || !ep->out) return -1;
  const clite_conv& c = *cv;
  if (c.stride != 1 || c.K % 64 || c.C % 8 || ep->ldc % 8) return -1;            // a 64-element K slab stays inside one (r, s)
  if (c.Ho != (c.H + 2 * c.pad - c.R) + 1 || c.Wo != (c.W + 2 * c.pad - c.S) + 1) return -1;
  // exactly the epilogue the bf16 backward issues for a unit inside a block: out = relu'(bits) * acc, (sum v, sum v (bn_y - mean)) into colsum
  if (!ep->relu_bits || !ep->bn_y || !ep->bn_stats || !ep->colsum || ep->dact_aux || ep->out_f32 || ep->alpha != 1.f || ep->residual || ep->mask_after_residual ||
      ep->atomic || ep->bias || ep->act || ep->preact || ep->drop_p > 0.f || ep->splitk_ws || ep->residual_subsample > 1 || ep->fp8_out || ep->fp8_amax || ep->fp8_scale)
    return -1;
  if ((size_t)c.N * c.Ho * c.Wo * c.K >= 0xF0000000ull || (size_t)c.N * c.H * c.W * c.C * 2 >= 0xF0000000ull) return -1;
  const int M = c.N * c.H * c.W, Ktot = c.R * c.S * c.K, ktiles = Ktot / 64;
  const uint32_t yb = (uint32_t)((size_t)c.N * c.Ho * c.Wo * c.K), wb = (uint32_t)((size_t)c.C * Ktot);
  typedef DmaKC<fp8, 128, 64, true> LA;
  typedef DmaKC<fp8, 128, 64, false> LB;
  LA la{dy8, yb, geom_dgrad8(c)};
  LB lb{wt8, wb, geom_dense(c.C, Ktot, Ktot)};
  RowMap rm{};
  const int tiles = ((M + 127) / 128) * ((c.C + 127) / 128);
  hipLaunchKernelGGL((igemm_dma_bn_kernel<bf16, F128, LA, LB, 1, false, true>), dim3(tiles), dim3(256), 0, (hipStream_t)stream, la, lb, *ep, rm, M, c.C, ktiles, 128,
                     dy_scales, w_scales);
  return (int)hipGetLastError();
}

extern "C" int clite_conv_fwd_fp8(const void* x8, const void* w8, const clite_conv* cv, const float* x_scales, const float* w_scales,
                                  const clite_epilogue* ep, void* stream) {
  if (!x8 || !w8 || !cv || !x_scales || !w_scales || check_ep8(ep, cv ? cv->K : 0)) return -1;
  const clite_conv& c = *cv;
  if (c.C % 16 || c.K % 8 || (c.R * c.S > 1 && c.C % 64)) return -1;            // a 64-element K slab must stay inside one (r, s)
  if (c.Ho != (c.H + 2 * c.pad - c.R) / c.stride + 1 || c.Wo != (c.W + 2 * c.pad - c.S) / c.stride + 1) return -1;
  if ((size_t)c.N * c.H * c.W * c.C >= 0xF0000000ull || (size_t)c.N * c.Ho * c.Wo * c.K * 2 >= 0xF0000000ull) return -1;
  const int M = c.N * c.Ho * c.Wo, Ktot = c.R * c.S * c.C;
  const uint32_t xb = (uint32_t)((size_t)c.N * c.H * c.W * c.C), wb = (uint32_t)((size_t)c.K * Ktot);
  return launch8(x8, xb, geom_fwd(c), w8, wb, geom_dense(c.K, Ktot, Ktot), *ep, x_scales, w_scales, M, c.K, Ktot, (hipStream_t)stream);
}
