// Grouped weight gradients: ONE launch for many independent dW_p[M_p][N_p] += A_p^T B_p (conv wgrad: A = dY, B = im2col(x); linear wgrad:
// A = dY, B = x), contracted over the pixels / tokens K_p.
//
// Why: a single weight gradient cannot fill the chip — its output is a few dozen 128 x 128 tiles — so igemm_dma_kernel splits K until ~500
// workgroups exist and pays for it: every split adds a whole f32 output tile of float-atomic traffic (chip-wide ~1.3 TB/s) plus a prologue
// and an epilogue. Measured on MI355X (tools/probe_tn_full.py): the BERT FFN gradient 3072 x 768 x 3840 runs at 328 TF/s alone, the same
// kernel at 12288 x 2304 x 3840 (tiles enough, no split) at 666 TF/s; 512 x 4608 x 6272 (layer4 3x3) 425 -> 716 TF/s. The weight gradients of a
// backward pass are independent of each other, so the captured train step (train_loop.TrainStep) collects them per stage and launches each
// collection once: thousands of workgroups without splitting the short-K problems at all; only the long-K / small-output ones (layer1, layer2:
// 100k-400k pixels into 64 x 576 outputs) are cut into k-chunks of <= 128 K tiles so that all workgroups run about equally long.
//
// Layout of the workspace the caller hands over (device copy + pinned host staging of the same size): [items][wg map]; the library fills the
// host side, copies it with one hipMemcpyAsync on the launch stream (a memcpy node when the stream is being captured) and launches. bf16 only;
// the exact-f32 mode and the deterministic-reduction mode run the members one by one through clite_conv_wgrad / clite_gemm_tn.
#include "igemm_wide.h"
#include "det.h"
#include <string.h>
#include <algorithm>
#include <vector>

using namespace clite;

namespace {

constexpr int GBK = 32;                 // K tile (bf16)
#ifndef CLITE_GROUP_KCHUNK
#define CLITE_GROUP_KCHUNK 256          // (the wave-simulator build of the tests sets 24, so that a member of 32 K tiles already runs in two chunks)
#endif
constexpr int KCHUNK = CLITE_GROUP_KCHUNK;             // K tiles per workgroup at most (8192 pixels / tokens)

template <class LA, class LB>
struct GroupItem {
  LA la;
  LB lb;
  float* out;
  int ldc, M, N, ktiles, chunk;
  int zeroed;          // CLITE_WGRAD_ZEROED: `out` is known to hold zeros on entry
  const float* row_scale;   // clite_wgrad_item.row_scale (ABI v12): per-output-row factor, or NULL
  const float* sa;          // kind 2 (fp8 operands): device {scale, 1 / scale} of a (dy, e5m2) ...
  const float* sb;          // ... and of b (x, e4m3); the product is de-quantised by sa[1] * sb[1] in the epilogue
};
struct WgEntry { uint32_t item, tile_m, tile_n, kchunk; };      // one per workgroup

template <class CFG, class LA, class LB, int NSTAGE>
__global__ __launch_bounds__(256) void igemm_group_kernel(const GroupItem<LA, LB>* __restrict__ items, const WgEntry* __restrict__ map) {
  typedef bf16 T;
  constexpr int BM = CFG::BM, BN = CFG::BN, BK = CFG::BK;
  constexpr int RM = CFG::RM, RN = CFG::RN;
  constexpr int STAGE = LA::BYTES + LB::BYTES;
  constexpr int LOADS_PER_TILE = LA::NI + LB::NI;
  __shared__ __attribute__((aligned(1024))) char smem[NSTAGE * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = wave_uniform(tid >> 6);
  const int wm0 = (wave / CFG::WAVES_N) * CFG::WM;
  const int wn0 = (wave % CFG::WAVES_N) * CFG::WN;
  const WgEntry e = map[blockIdx.x];
  if (wave_uniform((int)e.item) < 0) return;         // padding entry (Bucket::NO_ITEM): the XCD lists are of unequal length
  const GroupItem<LA, LB>& it = items[wave_uniform((int)e.item)];
  const LA la = it.la;
  const LB lb = it.lb;
  const int M = it.M, N = it.N;
  const int m0 = wave_uniform((int)e.tile_m) * BM, n0 = wave_uniform((int)e.tile_n) * BN;
  const int t_begin = wave_uniform((int)e.kchunk) * it.chunk;
  int t_end = t_begin + it.chunk;
  if (t_end > it.ktiles) t_end = it.ktiles;

  typename LA::State sa;
  typename LB::State sb;
  la.init(sa, m0, wave, lane, t_begin);
  lb.init(sb, n0, wave, lane, t_begin);

  f32x16 acc[RM][RN];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int aoff[RM][BK / 16], boff[RN][BK / 16];
#pragma unroll
  for (int ks = 0; ks < BK / 16; ++ks) {
#pragma unroll
    for (int i = 0; i < RM; ++i) aoff[i][ks] = LA::frag_off(wm0 + i * 32, ks, lane);
#pragma unroll
    for (int j = 0; j < RN; ++j) boff[j][ks] = LB::frag_off(wn0 + j * 32, ks, lane);
  }
#pragma unroll
  for (int pz = 0; pz < NSTAGE - 1; ++pz) {
    if (t_begin + pz < t_end) {
      DmaIssue<LA>::go(la, sa, smem + pz * STAGE, wave, lane, m0);
      DmaIssue<LB>::go(lb, sb, smem + pz * STAGE + LA::BYTES, wave, lane, n0);
    }
  }
  int buf = 0;
  for (int t = t_begin; t < t_end; ++t) {      // igemm_dma_kernel's pipeline: counted vmcnt, one raw barrier per K tile
    const int after = t_end - 1 - t;
    if (after >= 1) wait_vmcnt<LOADS_PER_TILE>();
    else wait_vmcnt<0>();
    barrier_raw();
    const char* abuf = smem + buf * STAGE;
    const char* bbuf = abuf + LA::BYTES;
    bf16x8 af0[RM], bf0[RN];
#pragma unroll
    for (int i = 0; i < RM; ++i) af0[i] = LA::frag_at(abuf + aoff[i][0]);
#pragma unroll
    for (int j = 0; j < RN; ++j) bf0[j] = LB::frag_at(bbuf + boff[j][0]);
    if (t + NSTAGE - 1 < t_end) {
      int nb = buf + NSTAGE - 1; if (nb >= NSTAGE) nb -= NSTAGE;
      DmaIssue<LA>::go(la, sa, smem + nb * STAGE, wave, lane, m0);
      DmaIssue<LB>::go(lb, sb, smem + nb * STAGE + LA::BYTES, wave, lane, n0);
    }
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 af[RM], bfr[RN];
#pragma unroll
      for (int i = 0; i < RM; ++i) af[i] = ks == 0 ? af0[i] : LA::frag_at(abuf + aoff[i][ks]);
#pragma unroll
      for (int j = 0; j < RN; ++j) bfr[j] = ks == 0 ? bf0[j] : LB::frag_at(bbuf + boff[j][ks]);
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) acc[i][j] = mfma32_bf16(af[i], bfr[j], acc[i][j]);
    }
    if (++buf == NSTAGE) buf = 0;
  }
  float* out = it.out;
  const int ldc = it.ldc;
  // A member whose contraction fits ONE chunk has exactly one workgroup per output element in this launch; when the caller also vouches that
  // the gradient buffer holds zeros (CLITE_WGRAD_ZEROED: the captured train step, whose update kernel leaves the arena zeroed and which visits
  // every weight once), `+=` is a plain store. Float atomics execute at the memory side at ~1.3 TB/s chip-wide — ~5 GB/s per CU — so the 256 KB
  // accumulator tile of a BERT weight gradient took ~50 us to drain, about as long as its main loop: 437 MB of atomic traffic per step for the
  // 109 M BERT parameters. (A plain read-modify-write instead of the store was measured SLOWER than the atomics: 15.29 vs 15.11 ms per step —
  // the loads put a round trip to HBM in front of every store.)
  const float* rs = it.row_scale;          // (workgroup-uniform: the folded BatchNorm backward's ka on a weight gradient contracted against dz)
#ifdef CLITE_GROUP_ATOMIC_ALWAYS          // A/B builds only
  const bool single = false;
#else
  const bool single = wave_uniform((int)(it.ktiles <= it.chunk && it.zeroed && !rs));          // (a row_scale member's correction terms are added from elsewhere: always +=)
#endif
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j) {
      const int col = n0 + wn0 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) {
          float* q = out + (size_t)row * ldc + col;
          const float v = rs ? acc[i][j][r] * rs[row] : acc[i][j][r];
          if (single) *q = v; else atomic_add_f32(q, v);
        }
      }
    }
}

// The same grouped launch on the 8-wave wide tiles of igemm_wide.h (K tile 64, one workgroup per CU): a 256 x 256 tile stages 256 B of
// operands per MFMA, 256 x 128 / 128 x 256 384 B, the 4-wave 128 x 128 x 32 tile 512 B — and the weight-gradient GEMMs (K = 3840 ... 25088
// pixels or tokens) are bound by exactly that L2 -> LDS traffic (500-620 TF/s on the narrow tiles). Members whose output
// has >= 256 rows and columns (every BERT matrix, the layer3 / layer4 convs) take the 256 x 256 tile; narrower outputs (layer1, layer2) stay on
// the 4-wave kernels above. Both operands are XC images (contraction index slow).
template <class CFG, class LA, class LB, int NSTAGE>
__global__ __launch_bounds__(512) void igemm_group_wide_kernel(const GroupItem<LA, LB>* __restrict__ items, const WgEntry* __restrict__ map) {
  constexpr int BM = CFG::BM, BN = CFG::BN;
  constexpr int RM = CFG::RM, RN = CFG::RN, KS = CFG::KS;
  constexpr int STAGE = LA::BYTES + LB::BYTES;
  constexpr int LOADS_PER_TILE = LA::NI + LB::NI;
  static_assert(CFG::KG == 1 && NSTAGE * STAGE <= 160 * 1024, "one k-group; LDS budget");
  __shared__ __attribute__((aligned(1024))) char smem[NSTAGE * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = wave_uniform(tid >> 6);
  const int wm0 = (wave / CFG::WAVES_N) * CFG::WM;
  const int wn0 = (wave % CFG::WAVES_N) * CFG::WN;
  const WgEntry e = map[blockIdx.x];
  if (wave_uniform((int)e.item) < 0) return;         // padding entry: the XCD lists are of unequal length
  const GroupItem<LA, LB>& it = items[wave_uniform((int)e.item)];
  const LA la = it.la;
  const LB lb = it.lb;
  const int M = it.M, N = it.N;
  const int m0 = wave_uniform((int)e.tile_m) * BM, n0 = wave_uniform((int)e.tile_n) * BN;
  const int t_begin = wave_uniform((int)e.kchunk) * it.chunk;
  int t_end = t_begin + it.chunk;
  if (t_end > it.ktiles) t_end = it.ktiles;

  typename LA::State sa;
  typename LB::State sb;
  la.init(sa, m0, wave, lane, t_begin);
  lb.init(sb, n0, wave, lane, t_begin);
  f32x16 acc[RM][RN];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int aoff[RM][KS], boff[RN][KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int i = 0; i < RM; ++i) aoff[i][ks] = LA::frag_off(wm0 + i * 32, ks, lane);
#pragma unroll
    for (int j = 0; j < RN; ++j) boff[j][ks] = LB::frag_off(wn0 + j * 32, ks, lane);
  }
#pragma unroll
  for (int pz = 0; pz < NSTAGE - 1; ++pz) {
    if (t_begin + pz < t_end) {
      WideIssue<LA>::go(la, sa, smem + pz * STAGE, wave, lane, m0);
      WideIssue<LB>::go(lb, sb, smem + pz * STAGE + LA::BYTES, wave, lane, n0);
    }
  }
  int buf = 0;
  for (int t = t_begin; t < t_end; ++t) {
    const int after = t_end - 1 - t;
    if (NSTAGE >= 3 && after >= 1) wait_vmcnt<LOADS_PER_TILE>();
    else wait_vmcnt<0>();
    barrier_raw();
    const char* abuf = smem + buf * STAGE;
    const char* bbuf = abuf + LA::BYTES;
    bf16x8 af0[RM], bf0[RN];
#pragma unroll
    for (int i = 0; i < RM; ++i) af0[i] = LA::frag_at(abuf + aoff[i][0]);
#pragma unroll
    for (int j = 0; j < RN; ++j) bf0[j] = LB::frag_at(bbuf + boff[j][0]);
    if (t + NSTAGE - 1 < t_end) {
      int nb = buf + NSTAGE - 1; if (nb >= NSTAGE) nb -= NSTAGE;
      WideIssue<LA>::go(la, sa, smem + nb * STAGE, wave, lane, m0);
      WideIssue<LB>::go(lb, sb, smem + nb * STAGE + LA::BYTES, wave, lane, n0);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 af[RM], bfr[RN];
#pragma unroll
      for (int i = 0; i < RM; ++i) af[i] = ks == 0 ? af0[i] : LA::frag_at(abuf + aoff[i][ks]);
#pragma unroll
      for (int j = 0; j < RN; ++j) bfr[j] = ks == 0 ? bf0[j] : LB::frag_at(bbuf + boff[j][ks]);
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) acc[i][j] = mfma32_bf16(af[i], bfr[j], acc[i][j]);
    }
    if (++buf == NSTAGE) buf = 0;
  }
  float* out = it.out;
  const int ldc = it.ldc;
  // A member whose contraction fits ONE chunk has exactly one workgroup per output element in this launch; when the caller also vouches that
  // the gradient buffer holds zeros (CLITE_WGRAD_ZEROED: the captured train step, whose update kernel leaves the arena zeroed and which visits
  // every weight once), `+=` is a plain store. Float atomics execute at the memory side at ~1.3 TB/s chip-wide — ~5 GB/s per CU — so the 256 KB
  // accumulator tile of a BERT weight gradient took ~50 us to drain, about as long as its main loop: 437 MB of atomic traffic per step for the
  // 109 M BERT parameters. (A plain read-modify-write instead of the store was measured SLOWER than the atomics: 15.29 vs 15.11 ms per step —
  // the loads put a round trip to HBM in front of every store.)
  const float* rs = it.row_scale;          // (workgroup-uniform: the folded BatchNorm backward's ka on a weight gradient contracted against dz)
#ifdef CLITE_GROUP_ATOMIC_ALWAYS          // A/B builds only
  const bool single = false;
#else
  const bool single = wave_uniform((int)(it.ktiles <= it.chunk && it.zeroed && !rs));          // (a row_scale member's correction terms are added from elsewhere: always +=)
#endif
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j) {
      const int col = n0 + wn0 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) {
          float* q = out + (size_t)row * ldc + col;
          const float v = rs ? acc[i][j][r] * rs[row] : acc[i][j][r];
          if (single) *q = v; else atomic_add_f32(q, v);
        }
      }
    }
}

// ---- fp8 members (ABI v12, CLITE_WGRAD_FP8: BASELINE configs[4]) ------------------------------------------------------------------------------
// dW += dq * dy8^T x8 with dy8 in OCP e5m2 (the copy clite_bn_bwd_apply's fused quantiser leaves for clite_conv_dgrad_fp8) and x8 in e4m3 (the copy
// clite_bn_apply leaves for clite_conv_fwd_fp8): the grouped 256 x 256 tile on v_mfma_scale_f32_32x32x64_f8f6f4 — a K tile of 64 pixels is ONE
// instruction per 32 x 32 block at twice the bf16 rate per k, and both operand images are half the bytes. Both operands are XC images (the contraction
// index, the pixel, is the slow one): [64 pixels][256 channels x 1 B]; the fragments come out of them through ds_read_b64_tr_b8 (intrin.h: lane s
// of a 16-lane group points at pixel row s >> 1, 8-byte column chunk s & 1, and lane i receives 8 consecutive pixels of channel i): four reads per
// operand and lane give 32 pixels — lanes 0..31 pixels 0..31 of the tile, lanes 32..63 pixels 32..63. The MFMA's nominal k order inside a lane differs
// (intrin.h: two runs of 16), but A and B are loaded by the same rule, and a contraction does not care in which order both operands enumerate k.
template <int ROWB> DEV int f8_tr_off(int x0, int lane) {
  const int s = lane & 15, g = lane >> 4;
  const int k = 32 * (g >> 1) + (s >> 1);                       // + 8 q for read q: same segment swizzle (k & 3 unchanged)
  const int xb = x0 + 16 * (g & 1) + 8 * (s & 1);               // byte column of this lane's 8-byte chunk
  return k * ROWB + ((((xb >> 4)) ^ (xc_seg_xor<ROWB>(k) << 2)) << 4) + (xb & 15);
}
template <class CFG, class LA, class LB, int NSTAGE>
__global__ __launch_bounds__(512) void igemm_group_fp8_kernel(const GroupItem<LA, LB>* __restrict__ items, const WgEntry* __restrict__ map) {
  constexpr int BM = CFG::BM, BN = CFG::BN;
  constexpr int RM = CFG::RM, RN = CFG::RN;
  constexpr int STAGE = LA::BYTES + LB::BYTES;
  constexpr int LOADS_PER_TILE = LA::NI + LB::NI;
  static_assert(CFG::KG == 1 && NSTAGE * STAGE <= 160 * 1024 && LA::ROWB == 256 && LB::ROWB == 256, "one k-group; LDS budget; 256-byte image rows");
  __shared__ __attribute__((aligned(1024))) char smem[NSTAGE * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = wave_uniform(tid >> 6);
  const int wm0 = (wave / CFG::WAVES_N) * CFG::WM;
  const int wn0 = (wave % CFG::WAVES_N) * CFG::WN;
  const WgEntry e = map[blockIdx.x];
  if (wave_uniform((int)e.item) < 0) return;
  const GroupItem<LA, LB>& it = items[wave_uniform((int)e.item)];
  const LA la = it.la;
  const LB lb = it.lb;
  const int M = it.M, N = it.N;
  const int m0 = wave_uniform((int)e.tile_m) * BM, n0 = wave_uniform((int)e.tile_n) * BN;
  const int t_begin = wave_uniform((int)e.kchunk) * it.chunk;
  int t_end = t_begin + it.chunk;
  if (t_end > it.ktiles) t_end = it.ktiles;

  typename LA::State sa;
  typename LB::State sb;
  la.init(sa, m0, wave, lane, t_begin);
  lb.init(sb, n0, wave, lane, t_begin);
  f32x16 acc[RM][RN];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int aoff[RM], boff[RN];
#pragma unroll
  for (int i = 0; i < RM; ++i) aoff[i] = f8_tr_off<LA::ROWB>(wm0 + i * 32, lane);
#pragma unroll
  for (int j = 0; j < RN; ++j) boff[j] = f8_tr_off<LB::ROWB>(wn0 + j * 32, lane);
#pragma unroll
  for (int pz = 0; pz < NSTAGE - 1; ++pz) {
    if (t_begin + pz < t_end) {
      WideIssue<LA>::go(la, sa, smem + pz * STAGE, wave, lane, m0);
      WideIssue<LB>::go(lb, sb, smem + pz * STAGE + LA::BYTES, wave, lane, n0);
    }
  }
  int buf = 0;
  for (int t = t_begin; t < t_end; ++t) {
    const int after = t_end - 1 - t;
    if (NSTAGE >= 3 && after >= 1) wait_vmcnt<LOADS_PER_TILE>();
    else wait_vmcnt<0>();
    barrier_raw();
    const char* abuf = smem + buf * STAGE;
    const char* bbuf = abuf + LA::BYTES;
    u32x4 alo[RM], ahi[RM], blo[RN], bhi[RN];
#pragma unroll
    for (int i = 0; i < RM; ++i) {
      const u32x2 q0 = lds_read_tr8(abuf + aoff[i]), q1 = lds_read_tr8(abuf + aoff[i] + 8 * LA::ROWB), q2 = lds_read_tr8(abuf + aoff[i] + 16 * LA::ROWB),
                  q3 = lds_read_tr8(abuf + aoff[i] + 24 * LA::ROWB);
      alo[i] = u32x4{q0[0], q0[1], q1[0], q1[1]};
      ahi[i] = u32x4{q2[0], q2[1], q3[0], q3[1]};
    }
#pragma unroll
    for (int j = 0; j < RN; ++j) {
      const u32x2 q0 = lds_read_tr8(bbuf + boff[j]), q1 = lds_read_tr8(bbuf + boff[j] + 8 * LB::ROWB), q2 = lds_read_tr8(bbuf + boff[j] + 16 * LB::ROWB),
                  q3 = lds_read_tr8(bbuf + boff[j] + 24 * LB::ROWB);
      blo[j] = u32x4{q0[0], q0[1], q1[0], q1[1]};
      bhi[j] = u32x4{q2[0], q2[1], q3[0], q3[1]};
    }
    if (t + NSTAGE - 1 < t_end) {
      int nb = buf + NSTAGE - 1; if (nb >= NSTAGE) nb -= NSTAGE;
      WideIssue<LA>::go(la, sa, smem + nb * STAGE, wave, lane, m0);
      WideIssue<LB>::go(lb, sb, smem + nb * STAGE + LA::BYTES, wave, lane, n0);
    }
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
      for (int j = 0; j < RN; ++j) acc[i][j] = mfma32x64_bf8_fp8(alo[i], ahi[i], blo[j], bhi[j], acc[i][j]);
    if (++buf == NSTAGE) buf = 0;
  }
  float* out = it.out;
  const int ldc = it.ldc;
  const float dq = it.sa[1] * it.sb[1];
  const float* rs = it.row_scale;
  const bool single = wave_uniform((int)(it.ktiles <= it.chunk && it.zeroed && !rs));
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j) {
      const int col = n0 + wn0 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) {
          float* q = out + (size_t)row * ldc + col;
          const float v = (rs ? acc[i][j][r] * rs[row] : acc[i][j][r]) * dq;
          if (single) *q = v; else atomic_add_f32(q, v);
        }
      }
    }
}

struct Plan { int M, N, Ktot, ktiles, chunk, nchunks, tm, tn; };
Plan plan(int M, int N, int Ktot, int BM, int BN, int bk = GBK, int kchunk = KCHUNK) {
  Plan p;
  p.M = M; p.N = N; p.Ktot = Ktot;
  p.ktiles = (Ktot + bk - 1) / bk;
  p.nchunks = (p.ktiles + kchunk - 1) / kchunk;
  p.chunk = (p.ktiles + p.nchunks - 1) / p.nchunks;
  p.nchunks = (p.ktiles + p.chunk - 1) / p.chunk;
  p.tm = (M + BM - 1) / BM; p.tn = (N + BN - 1) / BN;
  return p;
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

typedef TileCfg<128, 128, GBK, 64, 64> C128;
typedef TileCfg<64, 128, GBK, 64, 32> C64x128;
typedef TileCfg<128, 64, GBK, 32, 64> C128x64;
typedef DmaXCStrided<bf16, 128, GBK> XS128;
typedef DmaXCStrided<bf16, 64, GBK> XS64;
typedef DmaXCGather<bf16, 128, GBK> XG128;
typedef DmaXCGather<bf16, 64, GBK> XG64;
// 8-wave wide tile (K tile 64): 256 x 256 on a 2-stage ring (128 KB). The rectangular 256 x 128 / 128 x 256 tiles for layer2's 128-wide outputs
// were measured and dropped: that segment's group took 1.45 ms instead of 0.86 on the 4-wave tiles
typedef WideCfg<256, 256, 128, 64, 1> W256;
typedef DmaXCStrided<bf16, 256, WIDE_BK, WIDE_NW> WS256;
typedef DmaXCGather<bf16, 256, WIDE_BK, WIDE_NW> WG256;
typedef DmaXCStrided<uint8_t, 256, WIDE_BK, WIDE_NW> FS256;          // fp8 operand images: [64 pixels][256 channels x 1 B]
typedef DmaXCGather<uint8_t, 256, WIDE_BK, WIDE_NW> FG256;
constexpr int WKCHUNK = 128;            // K tiles of 64 per workgroup at most (8192 pixels / tokens, as KCHUNK)
template <class CFG> struct GroupStages { static constexpr int n = 3; };
template <> struct GroupStages<W256> { static constexpr int n = 2; };

// one bucket = one kernel instantiation; items and map entries are appended to the host staging image
template <class CFG, class LA, class LB, bool WIDE = false, bool F8 = false>
struct Bucket {
  std::vector<GroupItem<LA, LB>> items;
  std::vector<Plan> plans;
  std::vector<WgEntry> map;
  void add(const LA& la, const LB& lb, float* out, int ldc, const Plan& p, int zeroed, const float* row_scale = nullptr, const float* sa = nullptr,
           const float* sb = nullptr) {
    GroupItem<LA, LB> it;
    it.la = la; it.lb = lb; it.out = out; it.ldc = ldc; it.M = p.M; it.N = p.N; it.ktiles = p.ktiles; it.chunk = p.chunk; it.zeroed = zeroed;
    it.row_scale = row_scale;
    it.sa = sa; it.sb = sb;
    items.push_back(it);
    plans.push_back(p);
  }
  // Workgroup order = dispatch order, and workgroup i runs on XCD i % 8 (each XCD has its own 4 MB L2). All tiles of one (member, k-chunk)
  // read the same operand rows — the dy chunk is shared along a tile row, the x chunk along a tile column — so they are kept on ONE XCD as a
  // "bundle" (at most BUNDLE workgroups: about what an XCD runs at a time) and the bundles are dealt to the eight XCDs, longest K chunk first
  // and always to the least-loaded XCD (longest-processing-time-first keeps the tail of the launch short: a layer4 workgroup runs 196 K
  // tiles, a BERT one 120, a k-chunked layer1 one 256). Measured with FETCH_SIZE on MI355X: in plain grid order (one bundle spread over all
  // eight L2s) the grouped launches of a ResNet-50 + BERT step fetched 17.9 GB for 5.5 GB of operands.
  // Lists of unequal length are padded with NO_ITEM entries (the workgroup exits at once).
  static constexpr uint32_t NO_ITEM = 0xFFFFFFFFu;
  static constexpr int XCDS = 8, BUNDLE = WIDE ? 32 : 64;
  void build_map() {
    std::vector<uint32_t> order(items.size());
    for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return plans[a].chunk > plans[b].chunk; });
    std::vector<WgEntry> per_xcd[XCDS];
    uint64_t load[XCDS] = {};
    for (uint32_t idx : order) {
      const Plan& p = plans[idx];
      const int tiles = p.tm * p.tn;
      for (int c = 0; c < p.nchunks; ++c) {
        int len = p.chunk;
        if ((c + 1) * p.chunk > p.ktiles) len = p.ktiles - c * p.chunk;
        for (int t0 = 0; t0 < tiles; t0 += BUNDLE) {
          const int t1 = t0 + BUNDLE < tiles ? t0 + BUNDLE : tiles;
          int x = 0;
          for (int k = 1; k < XCDS; ++k)
            if (load[k] < load[x]) x = k;
          for (int t = t0; t < t1; ++t) per_xcd[x].push_back(WgEntry{idx, (uint32_t)(t / p.tn), (uint32_t)(t % p.tn), (uint32_t)c});
          load[x] += (uint64_t)(t1 - t0) * (uint64_t)len;
        }
      }
    }
    size_t longest = 0;
    for (int k = 0; k < XCDS; ++k) longest = per_xcd[k].size() > longest ? per_xcd[k].size() : longest;
    map.assign(longest * XCDS, WgEntry{NO_ITEM, 0, 0, 0});
    for (int k = 0; k < XCDS; ++k)
      for (size_t j = 0; j < per_xcd[k].size(); ++j) map[j * XCDS + k] = per_xcd[k][j];
  }
  size_t n_wgs() const { return map.size(); }        // after build_map()
  size_t item_bytes() const { return (items.size() * sizeof(GroupItem<LA, LB>) + 255) / 256 * 256; }
  size_t bytes() const { return item_bytes() + (n_wgs() * sizeof(WgEntry) + 255) / 256 * 256; }
  void stage(char* host, size_t off) {
    if (items.empty()) return;
    memcpy(host + off, items.data(), items.size() * sizeof(GroupItem<LA, LB>));
    memcpy(host + off + item_bytes(), map.data(), map.size() * sizeof(WgEntry));
  }
  int launch(char* dev, size_t off, hipStream_t st) const {
    if (items.empty()) return 0;
    if constexpr (F8)
      hipLaunchKernelGGL((igemm_group_fp8_kernel<CFG, LA, LB, 3>), dim3((unsigned)map.size()), dim3(512), 0, st,
                         (const GroupItem<LA, LB>*)(dev + off), (const WgEntry*)(dev + off + item_bytes()));
    else if constexpr (WIDE)
      hipLaunchKernelGGL((igemm_group_wide_kernel<CFG, LA, LB, GroupStages<CFG>::n>), dim3((unsigned)map.size()), dim3(512), 0, st,
                         (const GroupItem<LA, LB>*)(dev + off), (const WgEntry*)(dev + off + item_bytes()));
    else
      hipLaunchKernelGGL((igemm_group_kernel<CFG, LA, LB, 3>), dim3((unsigned)map.size()), dim3(256), 0, st,
                         (const GroupItem<LA, LB>*)(dev + off), (const WgEntry*)(dev + off + item_bytes()));
    return (int)hipGetLastError();
  }
};

bool fits32(size_t elems, size_t esize) { return elems * esize < 0xF0000000ull; }

}  // namespace

bool clite_group_wide_enabled();      // gemm_wide.hip: false under tile policy 4 (every launch on the 4-wave kernels)

extern "C" int clite_wgrad_group(int dtype, const clite_wgrad_item* items, int n, void* ws_dev, void* ws_host, uint64_t ws_bytes, void* stream) {
  if (!items || n <= 0) return n == 0 ? 0 : -1;
  hipStream_t st = (hipStream_t)stream;
  // members one by one: the exact-f32 mode (one k-ordered chain per output tracks the CPU reference), the deterministic-reduction mode
  // (one contribution per address) and callers without a workspace
  if (dtype != CLITE_BF16 || clite::deterministic() || !ws_dev || !ws_host) {
    for (int i = 0; i < n; ++i) {
      const clite_wgrad_item& w = items[i];
      int rc;
      if (w.row_scale || (w.kind & 0xFF) == 2) return -1;          // (grouped launches only)
      if ((w.kind & ~(CLITE_WGRAD_NARROW | CLITE_WGRAD_ZEROED | CLITE_WGRAD_SHORTK)) == 0) {
        rc = clite_conv_wgrad(w.a, w.b, &w.cv, w.out, stream);
      } else {
        clite_epilogue ep = {};
        ep.out = w.out; ep.ldc = w.ldc; ep.out_f32 = 1; ep.atomic = 1; ep.alpha = 1.f;
        rc = clite_gemm_tn(w.a, w.lda, w.b, w.ldb, w.M, w.N, w.K, dtype, &ep, stream);
      }
      if (rc) return rc;
    }
    return 0;
  }
  Bucket<C128, XS128, XG128> conv_full;
  Bucket<C64x128, XS64, XG128> conv_fewk;     // <= 64 output channels
  Bucket<C128x64, XS128, XG64> conv_fewc;     // <= 64 (r, s, ci) columns
  Bucket<C128, XS128, XS128> linear;
  Bucket<W256, WS256, WG256, true> conv_w256;             // >= 256 output channels and >= 256 columns
  Bucket<W256, WS256, WS256, true> linear_w256;
  Bucket<W256, FS256, FG256, true, true> conv_f8;          // fp8 operands (kind 2): the 256 x 256 tile on the block-scaled MFMA
  const bool wide_all = clite_group_wide_enabled();
  for (int i = 0; i < n; ++i) {
    const clite_wgrad_item& w = items[i];
    if (!w.a || !w.b || !w.out) return -1;
    const bool wide = wide_all && !(w.kind & CLITE_WGRAD_NARROW);
    const int zeroed = (w.kind & CLITE_WGRAD_ZEROED) ? 1 : 0;
    const int kind = w.kind & ~(CLITE_WGRAD_NARROW | CLITE_WGRAD_ZEROED | CLITE_WGRAD_SHORTK);
    const int kc = (w.kind & CLITE_WGRAD_SHORTK) ? KCHUNK / 4 : KCHUNK, wkc = (w.kind & CLITE_WGRAD_SHORTK) ? WKCHUNK / 4 : WKCHUNK;
    if (kind == 0) {
      const clite_conv& c = w.cv;
      if (c.dtype != CLITE_BF16 || c.C % 8 || c.K % 8 || (c.R * c.S > 1 && (c.C % 32 || c.K % 32))) return -1;
      if (!fits32((size_t)c.N * c.H * c.W * c.C, 4) || !fits32((size_t)c.N * c.Ho * c.Wo * c.K, 4)) return -1;
      const int P = c.N * c.Ho * c.Wo, Ncols = c.R * c.S * c.C;
      const uint32_t yb = (uint32_t)((size_t)P * c.K * 2), xb = (uint32_t)((size_t)c.N * c.H * c.W * c.C * 2);
      if (wide && c.K >= 256 && Ncols >= 256) conv_w256.add(WS256{w.a, yb, c.K, c.K, P, 1}, WG256{w.b, xb, geom_fwd(c)}, w.out, Ncols, plan(c.K, Ncols, P, 256, 256, WIDE_BK, wkc), zeroed, w.row_scale);
      else if (c.K <= 64) conv_fewk.add(XS64{w.a, yb, c.K, c.K, P, 1}, XG128{w.b, xb, geom_fwd(c)}, w.out, Ncols, plan(c.K, Ncols, P, 64, 128, GBK, kc), zeroed, w.row_scale);
      else if (Ncols <= 64) conv_fewc.add(XS128{w.a, yb, c.K, c.K, P, 1}, XG64{w.b, xb, geom_fwd(c)}, w.out, Ncols, plan(c.K, Ncols, P, 128, 64, GBK, kc), zeroed, w.row_scale);
      else conv_full.add(XS128{w.a, yb, c.K, c.K, P, 1}, XG128{w.b, xb, geom_fwd(c)}, w.out, Ncols, plan(c.K, Ncols, P, 128, 128, GBK, kc), zeroed, w.row_scale);
    } else if (kind == 2) {
      // conv weight gradient on fp8 operands: a = dy8 [N][Ho][Wo][K] e5m2, b = x8 [N][H][W][C] e4m3, a_scales / b_scales their {scale, 1 / scale}
      const clite_conv& c = w.cv;
      if (!w.a_scales || !w.b_scales || c.dtype != CLITE_BF16 || c.C % 16 || c.K % 16 || (c.R * c.S > 1 && (c.C % 32 || c.K % 32))) return -1;
      if (!fits32((size_t)c.N * c.H * c.W * c.C, 1) || !fits32((size_t)c.N * c.Ho * c.Wo * c.K, 1)) return -1;
      const int P = c.N * c.Ho * c.Wo, Ncols = c.R * c.S * c.C;
      const uint32_t yb = (uint32_t)((size_t)P * c.K), xb = (uint32_t)((size_t)c.N * c.H * c.W * c.C);
      conv_f8.add(FS256{w.a, yb, c.K, c.K, P, 1}, FG256{w.b, xb, geom_fwd(c)}, w.out, Ncols, plan(c.K, Ncols, P, 256, 256, WIDE_BK, wkc), zeroed, w.row_scale, w.a_scales,
                  w.b_scales);
    } else if (kind == 1) {
      if (w.M <= 0 || w.N <= 0 || w.K <= 0 || w.M % 8 || w.N % 8 || w.lda % 8 || w.ldb % 8 || w.lda < w.M || w.ldb < w.N) return -1;
      if (!fits32((size_t)w.K * w.lda, 4) || !fits32((size_t)w.K * w.ldb, 4)) return -1;
      const uint32_t ab = (uint32_t)((((size_t)w.K - 1) * w.lda + w.M) * 2), bb = (uint32_t)((((size_t)w.K - 1) * w.ldb + w.N) * 2);
      if (wide && w.M >= 256 && w.N >= 256) linear_w256.add(WS256{w.a, ab, w.lda, w.M, w.K, 1}, WS256{w.b, bb, w.ldb, w.N, w.K, 1}, w.out, w.ldc, plan(w.M, w.N, w.K, 256, 256, WIDE_BK, wkc), zeroed, w.row_scale);
      else linear.add(XS128{w.a, ab, w.lda, w.M, w.K, 1}, XS128{w.b, bb, w.ldb, w.N, w.K, 1}, w.out, w.ldc, plan(w.M, w.N, w.K, 128, 128, GBK, kc), zeroed, w.row_scale);
    } else {
      return -1;
    }
  }
  conv_full.build_map(); conv_fewk.build_map(); conv_fewc.build_map(); linear.build_map();
  conv_w256.build_map(); linear_w256.build_map(); conv_f8.build_map();
  const size_t o0 = 0, o1 = o0 + conv_full.bytes(), o2 = o1 + conv_fewk.bytes(), o3 = o2 + conv_fewc.bytes(), o4 = o3 + linear.bytes(),
               o7 = o4 + conv_w256.bytes(), o8 = o7 + linear_w256.bytes(), need = o8 + conv_f8.bytes();
  if (need > ws_bytes) return -2;
  char* host = (char*)ws_host;
  char* dev = (char*)ws_dev;
  // fill the host image, copy it once, then enqueue the (up to six) launches: they read `dev` after the copy, in stream order. The long
  // one-per-CU workgroups of the wide buckets go first; the 4-wave buckets fill in behind them
  conv_full.stage(host, o0); conv_fewk.stage(host, o1); conv_fewc.stage(host, o2); linear.stage(host, o3);
  conv_w256.stage(host, o4); linear_w256.stage(host, o7); conv_f8.stage(host, o8);
  int rc = (int)hipMemcpyAsync(dev, host, need, hipMemcpyHostToDevice, st);
  if (rc) return rc;
  if ((rc = conv_f8.launch(dev, o8, st))) return rc;
  if ((rc = conv_w256.launch(dev, o4, st))) return rc;
  if ((rc = linear_w256.launch(dev, o7, st))) return rc;
  if ((rc = conv_full.launch(dev, o0, st))) return rc;
  if ((rc = conv_fewk.launch(dev, o1, st))) return rc;
  if ((rc = conv_fewc.launch(dev, o2, st))) return rc;
  return linear.launch(dev, o3, st);
}

// Workspace bytes that certainly hold the descriptors of `n_items` members with `total_workgroups` workgroups in all (each member contributes
// ceil(M/BM) * ceil(N/BN) * ceil(ceil(K/32)/KCHUNK) of them, KCHUNK = 256): 512 B per member, 16 B per map entry, alignment slack.
// build_map() balances the eight per-XCD lists by WORK (tiles x K length), not by entry count, and pads every list to the longest one: a
// bucket that mixes long-K and short-K members can hold several times its real entries. The longest list cannot exceed the real entry
// count, so 8 x total_workgroups entries is a bound that no mix of members breaks; it is 16 B per entry, i.e. about 1 MB for the
// largest step this build runs.
extern "C" int clite_wgrad_group_workspace(int n_items, int64_t total_workgroups, uint64_t* bytes) {
  if (n_items < 0 || total_workgroups < 0 || !bytes) return -1;
  *bytes = (uint64_t)n_items * 512 + (uint64_t)total_workgroups * 16 * 8 + 4 * 4096;
  return 0;
}
