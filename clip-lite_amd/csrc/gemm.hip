// Host launchers (C ABI, include/clite.h) for the implicit-GEMM engine in igemm.h.
#include "igemm_dma.h"
#include "wide_api.h"
#include "det.h"
#include <type_traits>
#include <atomic>
#include <stdlib.h>

using namespace clite;

static std::atomic<int> g_deterministic{0};
bool clite::deterministic() { return g_deterministic.load(std::memory_order_relaxed) != 0; }

namespace {

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

ConvGeom geom_dense(int rows, int K, int ld = 0) {  // [rows][K] (row stride ld) seen as a 1x1 window over a rows x 1 x 1 image
  ConvGeom g;
  g.H = 1; g.W = 1; g.C = K;
  g.sN = ld ? ld : K; g.sH = 0; g.sW = 0;
  g.RH = 1; g.RW = 1; g.R = 1; g.S = 1; g.stride = 1; g.pad = 0; g.padw = 0;
  g.rows = rows; g.concat = 0;
  g.div_hw = fastdiv_make(1);
  g.div_w = fastdiv_make(1);
  return g;
}

bool fits32(size_t elems, size_t esize) { return elems * esize < 0xF0000000ull; }

template <typename T> struct Cfg {
  static constexpr int BK = 64 / (int)sizeof(T);        // 64 bytes of K per tile row: bf16 -> 32, f32 -> 16
  typedef TileCfg<128, 128, BK, 64, 64> C128;
  typedef TileCfg<256, 64, BK, 64, 64> C256x64;
  typedef TileCfg<64, 128, BK, 64, 32> C64x128;     // weight gradients of convs with <= 64 output channels
  typedef TileCfg<128, 64, BK, 32, 64> C128x64;     // ... or <= 64 (r, s, ci) columns
};

// Kernel-selection knobs. The product library has none: every choice below is a fixed function of the problem shape. A diagnostic build
// (-DCLITE_DIAG=1, tools/README.md) reads them from the environment for A/B timing and also carries the round-1 register-staged engine.
#ifndef CLITE_DIAG
#define CLITE_DIAG 0
#endif
#ifndef CLITE_BN_SLOTS
#define CLITE_BN_SLOTS 512      // resident workgroups of igemm_dma_bn_kernel on the chip: 2 per CU x 256 CUs (the wave-simulator build of the tests sets 4)
#endif
#if CLITE_DIAG
int diag_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
bool use_dma() { static int v = -1; if (v < 0) v = diag_env("CLITE_IGEMM_LEGACY", 0) ? 0 : 1; return v == 1; }
int stages_pref() { static int v = -1; if (v < 0) v = diag_env("CLITE_IGEMM_STAGES", 0); return v; }     // 3|4 forces a ring depth
int g2_pref() { static int v = -1; if (v < 0) v = diag_env("CLITE_IGEMM_G2", 1); return v; }              // 0 off, 2 forces the two-K-group kernel
int xcd_split_pref() { static int v = -1; if (v < 0) v = diag_env("CLITE_XCD_SPLIT", 1); return v; }
int split_target_pref() { static int v = -1; if (v < 0) v = diag_env("CLITE_SPLIT_TARGET", 0); return v; }
int splitk_ws_pref() { static int v = -1; if (v < 0) v = diag_env("CLITE_SPLITK_WS", 1); return v; }
#else
constexpr bool use_dma() { return true; }
constexpr int stages_pref() { return 0; }
constexpr int g2_pref() { return 1; }
constexpr int xcd_split_pref() { return 1; }
constexpr int split_target_pref() { return 0; }
constexpr int splitk_ws_pref() { return 1; }
#endif
// operands as gemm_wide.hip's 8-wave kernels take them (wide_api.h)
template <typename T, int ROWS, int BK, bool D> WideOperand wide_op(const GatherKC<T, ROWS, BK, D>& l) {
  WideOperand o{};
  o.kind = D ? WOP_KC_DGRAD : WOP_KC; o.ptr = l.ptr; o.bytes = l.bytes; o.g = l.g;
  return o;
}
template <typename T, int COLS, int BK> WideOperand wide_op(const StridedXC<T, COLS, BK>& l) {
  WideOperand o{};
  o.kind = WOP_XC_STRIDED; o.ptr = l.ptr; o.bytes = l.bytes; o.ld = l.ld; o.Cx = l.Cx; o.Ck = l.Ck; o.RS = l.RS;
  return o;
}
template <typename T, int COLS, int BK> WideOperand wide_op(const GatherXC<T, COLS, BK>& l) {
  WideOperand o{};
  o.kind = WOP_XC_GATHER; o.ptr = l.ptr; o.bytes = l.bytes; o.g = l.g;
  return o;
}
template <class L> struct IsDgrad { static constexpr bool value = false; };
template <typename T, int ROWS, int BK> struct IsDgrad<GatherKC<T, ROWS, BK, true>> { static constexpr bool value = true; };
template <class L> struct ToDma;
template <typename T, int ROWS, int BK, bool D> struct ToDma<GatherKC<T, ROWS, BK, D>> {
  typedef DmaKC<T, ROWS, BK, D> type;
  static type make(const GatherKC<T, ROWS, BK, D>& l) { return type{l.ptr, l.bytes, l.g}; }
};
template <typename T, int COLS, int BK> struct ToDma<StridedXC<T, COLS, BK>> {
  typedef DmaXCStrided<T, COLS, BK> type;
  static type make(const StridedXC<T, COLS, BK>& l) { return type{l.ptr, l.bytes, l.ld, l.Cx, l.Ck, l.RS}; }
};
template <typename T, int COLS, int BK> struct ToDma<GatherXC<T, COLS, BK>> {
  typedef DmaXCGather<T, COLS, BK> type;
  static type make(const GatherXC<T, COLS, BK>& l) { return type{l.ptr, l.bytes, l.g}; }
};

// ---- deterministic-reduction mode (det.h): column statistics of a GEMM output, from the stored tensor ---------------------------
// Same quantities as the fused epilogues accumulate (sum v, sum v^2 of the stored values; or sum v, sum v*(bn_y - mean) when bn_y is set),
// but each (replica, column) address receives exactly one contribution: row slab s of the GEMM rows goes to replica s, the 8 row lanes of
// a workgroup fold through LDS in lane order, and a thread walks its rows in increasing order.
template <typename T>
__global__ __launch_bounds__(256) void colstats_det_kernel(clite_epilogue ep, RowMap rm, int M, int N, int rows_per_slab) {
  __shared__ float red[8][32 * 16];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int col = (blockIdx.x * 32 + cx) * 8;
  const bool cvalid = col < N;
  float mean[8], s1[8], s2[8];
  zero8(mean); zero8(s1); zero8(s2);
  if (cvalid && ep.bn_y) {
    for (int r = 0; r < ep.bn_replicas; ++r)
#pragma unroll
      for (int e = 0; e < 8; ++e) mean[e] += ep.bn_stats[(size_t)r * ep.bn_rstride + col + e];
#pragma unroll
    for (int e = 0; e < 8; ++e) mean[e] *= ep.bn_inv_count;
  }
  int row_begin = blockIdx.y * rows_per_slab, row_end = row_begin + rows_per_slab;
  if (row_end > M) row_end = M;
  if (cvalid) {
    for (int r = row_begin + ry; r < row_end; r += 8) {
      const size_t idx = map_row(rm, r) * ep.ldc + col;
      float v[8];
      if (ep.out_f32 || sizeof(T) == 4) load8((const float*)ep.out + idx, v); else load8((const T*)ep.out + idx, v);
      if (ep.bn_y) {
        float y[8];
        load8((const T*)ep.bn_y + idx, y);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * (y[e] - mean[e]); }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[ry][cx * 16 + e] = s1[e]; red[ry][cx * 16 + 8 + e] = s2[e]; }
  __syncthreads();
  float* crep = ep.colsum + (ep.colsum_replicas > 1 ? (size_t)blockIdx.y * ep.colsum_stride : 0);
  for (int j = threadIdx.x; j < 32 * 16; j += 256) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += red[r][j];
    const int e = j & 15, c = (blockIdx.x * 32 + (j >> 4)) * 8 + (e & 7);
    if (c < N && (e < 8 || ep.colsum_rows != 1)) atomic_add_f32(crep + (e >= 8 ? N : 0) + c, s);      // the only contribution to this address in this launch
  }
}
template <typename T>
int colstats_det(const clite_epilogue& ep, const RowMap& rm, int M, int N, hipStream_t st) {
  int slabs = ep.colsum_replicas > 1 ? ep.colsum_replicas : 1;
  if (slabs > M) slabs = M;
  int rps = (M + slabs - 1) / slabs;
  slabs = (M + rps - 1) / rps;
  hipLaunchKernelGGL(colstats_det_kernel<T>, dim3((N / 8 + 31) / 32, slabs), dim3(256), 0, st, ep, rm, M, N, rps);
  return (int)hipGetLastError();
}

// Split-bf16 form of the exact-f32 mode's MFMA loops (igemm_dma.h split_bf16x3): process-wide switch like the deterministic mode
static std::atomic<int> g_f32_split{0};
#ifdef CLITE_PROBE_AFRAG
// probe builds (tools/probe_afrag.py): scale / shift of a BatchNorm applied to the A fragments of the next persistent 1 x 1 forward launches
static const float* g_probe_aff[2] = {nullptr, nullptr};
extern "C" void clite_probe_set_affine(const float* scale, const float* shift) { g_probe_aff[0] = scale; g_probe_aff[1] = shift; }
#endif
bool f32_split() { return g_f32_split.load(std::memory_order_relaxed) != 0; }

template <typename T, class CFG, class LA, class LB>
int launch(const LA& la, const LB& lb, const clite_epilogue& ep, int M, int N, int Ktot, int splits, hipStream_t st, RowMap rm = RowMap{}) {
  if (deterministic()) {
    if (ep.colsum) {       // statistics from the stored tensor, one contribution per address (colstats_det_kernel)
      clite_epilogue e2 = ep;
      e2.colsum = nullptr;
      int rc = launch<T, CFG>(la, lb, e2, M, N, Ktot, splits, st, rm);
      return rc ? rc : colstats_det<T>(ep, rm, M, N, st);
    }
    splits = 1;            // one workgroup per output tile over the whole K range: a single add per address
  }
  if constexpr (sizeof(T) == 2 && std::is_same<CFG, typename Cfg<T>::C128>::value) {
    // bf16, 128-column-or-wider outputs: the wide-K 8-wave kernels (igemm_wide.h) where the shape allows
    if (use_dma()) {
      const int rcw = launch_wide(wide_op(la), wide_op(lb), ep, rm, M, N, Ktot, splits, st);
      if (rcw != WIDE_NOT_TAKEN) return rcw;
    }
  }
  int ktiles = (Ktot + CFG::BK - 1) / CFG::BK;
  if (splits < 1) splits = 1;
  if (splits > ktiles) splits = ktiles;
  int per = (ktiles + splits - 1) / splits;
  splits = (ktiles + per - 1) / per;
  int tiles = ((M + CFG::BM - 1) / CFG::BM) * ((N + CFG::BN - 1) / CFG::BN);
  // >= 8 K splits: 1-D grid in which each XCD owns whole splits (igemm_dma.h)
  const int xsplits = (xcd_split_pref() && splits >= 8 && use_dma()) ? splits : 0;
  const dim3 grid = xsplits ? dim3(8 * ((splits + 7) / 8) * tiles, 1, 1) : dim3(tiles, 1, splits);
  if (use_dma()) {
    typedef typename ToDma<LA>::type DA;
    typedef typename ToDma<LB>::type DB;
    // a grid that gives each CU at most ~2 workgroups hides DMA latency with a deeper ring instead (4 stages x 16 KB = 64 KB)
    constexpr int STAGE = DA::BYTES + DB::BYTES;
    if (ep.bn_y || ep.mask_after_residual) {     // BatchNorm-backward epilogue: its own (register-heavier) instantiation, dgrad loaders only
      // it implements exactly: out = [(alpha*acc) (* relu'(aux))] (+ residual) [(* relu'(aux))], colsum; nothing else
      if (ep.atomic || ep.act || ep.preact || ep.drop_p > 0.f || (ep.dact_aux && ep.dact != 1) || (ep.dact_aux && ep.relu_bits) || splits != 1) return -1;
      if constexpr (IsDgrad<LA>::value) {
        // the two forms of the bf16 ResNet backward get their own instantiations (igemm_epilogue_bn's FORM); everything else the run-time form
        constexpr bool kc_b = !LB::XC;
        int form = 0;
        if (sizeof(T) == 2 && kc_b && ep.relu_bits && ep.bn_y && !ep.dact_aux && !ep.out_f32 && ep.alpha == 1.f) {
          if (!ep.residual && !ep.mask_after_residual) form = ep.bias ? 5 : 1;
          else if (ep.residual && ep.mask_after_residual) form = rm.on == 2 ? 3 : 2;
        }
        // row-range persistent form (igemm_dma_bn_kernel) for the short-K (1 x 1) dgrads whose tile count exceeds the resident slots — two
        // workgroups per CU, three for the specialised forms on the 128 x 128 tile (half-tile staging: 48 KB of LDS, 168 registers); one
        // tile per workgroup otherwise
        const int slots = (CLITE_BN_HALF && form && CFG::BM == 128) ? CLITE_BN_SLOTS * 3 / 2 : CLITE_BN_SLOTS;
        const int tiles_n = (N + CFG::BN - 1) / CFG::BN, tiles_m = (M + CFG::BM - 1) / CFG::BM;
        int rows_per_wg = CFG::BM, slices = tiles_m;
        if ((la.g.R * la.g.S == 1 || la.g.concat) && (long)tiles_m * tiles_n > slots && tiles_n <= slots && !deterministic()) {
          slices = slots / tiles_n;
          rows_per_wg = ((M + slices - 1) / slices + 7) & ~7;
          if (rows_per_wg < CFG::BM) rows_per_wg = CFG::BM;
          slices = (M + rows_per_wg - 1) / rows_per_wg;
        }
        const dim3 g(slices * tiles_n);
        if constexpr (sizeof(T) == 2 && kc_b) {
          if (form == 1) hipLaunchKernelGGL((igemm_dma_bn_kernel<T, CFG, DA, DB, 1>), g, dim3(256), 0, st, ToDma<LA>::make(la), ToDma<LB>::make(lb), ep, rm, M, N, ktiles, rows_per_wg, nullptr, nullptr);
          else if (form == 2) hipLaunchKernelGGL((igemm_dma_bn_kernel<T, CFG, DA, DB, 2>), g, dim3(256), 0, st, ToDma<LA>::make(la), ToDma<LB>::make(lb), ep, rm, M, N, ktiles, rows_per_wg, nullptr, nullptr);
          else if (form == 3) hipLaunchKernelGGL((igemm_dma_bn_kernel<T, CFG, DA, DB, 3>), g, dim3(256), 0, st, ToDma<LA>::make(la), ToDma<LB>::make(lb), ep, rm, M, N, ktiles, rows_per_wg, nullptr, nullptr);
          else if (form == 5) hipLaunchKernelGGL((igemm_dma_bn_kernel<T, CFG, DA, DB, 5>), g, dim3(256), 0, st, ToDma<LA>::make(la), ToDma<LB>::make(lb), ep, rm, M, N, ktiles, rows_per_wg, nullptr, nullptr);
          else hipLaunchKernelGGL((igemm_dma_bn_kernel<T, CFG, DA, DB, 0>), g, dim3(256), 0, st, ToDma<LA>::make(la), ToDma<LB>::make(lb), ep, rm, M, N, ktiles, rows_per_wg, nullptr, nullptr);
        } else {
          bool done = false;
          if constexpr (sizeof(T) == 4) {
            if (f32_split()) {
              hipLaunchKernelGGL((igemm_dma_bn_kernel<T, CFG, DA, DB, 0, true>), g, dim3(256), 0, st, ToDma<LA>::make(la), ToDma<LB>::make(lb), ep, rm, M, N, ktiles, rows_per_wg, nullptr, nullptr);
              done = true;
            }
          }
          if (!done)
            hipLaunchKernelGGL((igemm_dma_bn_kernel<T, CFG, DA, DB, 0>), g, dim3(256), 0, st, ToDma<LA>::make(la), ToDma<LB>::make(lb), ep, rm, M, N, ktiles, rows_per_wg, nullptr, nullptr);
        }
        return (int)hipGetLastError();
      } else {
        return -1;
      }
    }
    // The HBM-bound 1 x 1 FORWARD convolutions of the 56 x 56 / 28 x 28 stages (K <= 512, thousands of tiles, plain bf16 store + column statistics:
    // 3.4 - 3.8 TB/s as one tile per workgroup) in the row-range persistent structure of the BatchNorm-backward dgrads, which
    // sustains 5.2 TB/s on the same tensors (igemm_epilogue_bn FORM 4: no operands, sums of v and v^2): every workgroup is resident at once, walks
    // its own row range of one column tile and adds its column sums once.
#ifndef CLITE_NO_PERSIST_FWD
    if constexpr (sizeof(T) == 2 && !LA::XC && !LB::XC && !IsDgrad<LA>::value) {
      const bool plain_fwd = !ep.atomic && !ep.out_f32 && !ep.preact && ep.act == CLITE_ACT_NONE && !ep.dact_aux && ep.drop_p <= 0.f && !ep.residual && !ep.bias &&
                             ep.alpha == 1.f && !ep.relu_bits && !ep.splitk_ws && splits == 1 && rm.on == 0;
      const int tiles_n = (N + CFG::BN - 1) / CFG::BN, tiles_m = (M + CFG::BM - 1) / CFG::BM;
      if (plain_fwd && la.g.R * la.g.S == 1 && ktiles <= 16 && (long)tiles_m * tiles_n > 2 * CLITE_BN_SLOTS && tiles_n <= CLITE_BN_SLOTS && !deterministic() &&
          tile_policy_value() == 0) {
        int slices = CLITE_BN_SLOTS / tiles_n;
        int rows_per_wg = ((M + slices - 1) / slices + 7) & ~7;
        if (rows_per_wg < CFG::BM) rows_per_wg = CFG::BM;
        slices = (M + rows_per_wg - 1) / rows_per_wg;
#ifdef CLITE_PROBE_AFRAG
        if (g_probe_aff[0] && ktiles * CFG::BK <= 512) {
          hipLaunchKernelGGL((igemm_dma_bn_kernel<T, CFG, DA, DB, 4, false, false, true>), dim3(slices * tiles_n), dim3(256), 0, st, ToDma<LA>::make(la), ToDma<LB>::make(lb), ep,
                             rm, M, N, ktiles, rows_per_wg, g_probe_aff[0], g_probe_aff[1]);
          return (int)hipGetLastError();
        }
#endif
        hipLaunchKernelGGL((igemm_dma_bn_kernel<T, CFG, DA, DB, 4>), dim3(slices * tiles_n), dim3(256), 0, st, ToDma<LA>::make(la), ToDma<LB>::make(lb), ep, rm, M, N, ktiles,
                           rows_per_wg, nullptr, nullptr);
        return (int)hipGetLastError();
      }
    }
#endif
    // at most one workgroup per CU: two K-groups per workgroup (8 waves, in-workgroup split-K) instead of one wave per SIMD. bf16 only:
    // the exact-f32 parity mode keeps one k-ordered fmaf chain per output, which is what tracks the CPU reference most closely
    if (g2_pref() != 0 && 6 * STAGE <= 160 * 1024 && (((long)tiles * splits <= 256 && per >= 8 && sizeof(T) == 2) || g2_pref() == 2)) {
      // (4- and 5-stage rings for this kernel were measured and dropped: 38.6 / 39.6 / 42.6 us on the N = 768 BERT shapes)
      hipLaunchKernelGGL((igemm_dma_kernel_g2<T, CFG, DA, DB, 3>), grid, dim3(512), 0, st, ToDma<LA>::make(la), ToDma<LB>::make(lb), ep, rm,
                         M, N, ktiles, per, xsplits);
      return (int)hipGetLastError();
    }
    if constexpr (sizeof(T) == 4) {
      if (f32_split()) {          // f32 storage, three bf16 MFMAs per product (clite_set_f32_split)
        if (4 * STAGE <= 65536 && (long)tiles * splits <= 2 * 256 && per >= 4)
          hipLaunchKernelGGL((igemm_dma_kernel<T, CFG, DA, DB, 4, 0, true>), grid, dim3(256), 0, st, ToDma<LA>::make(la), ToDma<LB>::make(lb), ep, rm, M, N, ktiles, per, xsplits);
        else
          hipLaunchKernelGGL((igemm_dma_kernel<T, CFG, DA, DB, 3, 0, true>), grid, dim3(256), 0, st, ToDma<LA>::make(la), ToDma<LB>::make(lb), ep, rm, M, N, ktiles, per, xsplits);
        return (int)hipGetLastError();
      }
    }
    bool deep = 4 * STAGE <= 65536 && (long)tiles * splits <= 2 * 256 && per >= 4 && stages_pref() != 3;
    if (deep || (stages_pref() == 4 && 4 * STAGE <= 65536))
      hipLaunchKernelGGL((igemm_dma_kernel<T, CFG, DA, DB, 4>), grid, dim3(256), 0, st, ToDma<LA>::make(la), ToDma<LB>::make(lb), ep, rm,
                         M, N, ktiles, per, xsplits);
    else if (sizeof(T) == 2 && !ep.atomic && !ep.out_f32 && !ep.preact && ep.act == CLITE_ACT_NONE && !ep.dact_aux && ep.drop_p <= 0.f && !ep.residual && !ep.bn_y &&
             !ep.mask_after_residual)
      // plain bf16 store (+ bias, + column statistics): the branch-free epilogue instantiation
      hipLaunchKernelGGL((igemm_dma_kernel<T, CFG, DA, DB, 3, 2>), grid, dim3(256), 0, st, ToDma<LA>::make(la), ToDma<LB>::make(lb), ep, rm,
                         M, N, ktiles, per, xsplits);
    else
      hipLaunchKernelGGL((igemm_dma_kernel<T, CFG, DA, DB, 3>), grid, dim3(256), 0, st, ToDma<LA>::make(la), ToDma<LB>::make(lb), ep, rm,
                         M, N, ktiles, per, xsplits);
  } else {
#if CLITE_DIAG
    hipLaunchKernelGGL((igemm_kernel<T, CFG, LA, LB>), dim3(tiles, 1, splits), dim3(256), 0, st, la, lb, ep, rm, M, N, ktiles, per);
#endif
  }
  return (int)hipGetLastError();
}

// Split-K factor for float-atomic accumulation (weight gradients). Every split adds a full output tile of atomic traffic
// (chip-wide ~1.3 TB/s) plus a prologue/epilogue, so split only as far as needed to give every CU about two workgroups, and
// never below 8 K tiles per split.
int pick_splits(int M, int N, int ktiles, int BM = 128, int BN = 128, bool window = false) {
  long tiles = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  const int target = split_target_pref();
  // measured inside the step (tools/layer_profile.py): dense / 1x1 weight gradients with few output tiles are fastest at ~1
  // workgroup per CU (BERT 768x768: 39 -> 32 us; the 1x1 convs 5-15 % faster), windowed (3x3, 7x7) ones and larger outputs at ~2
  long tgt = target > 0 ? target : ((tiles < 64 && !window) ? 256 : 512);
  long want = (tgt + tiles - 1) / tiles;
  long cap = ktiles / 8;
  if (cap < 1) cap = 1;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  return (int)want;
}

int check_ep(const clite_epilogue* ep, int N) {
  if (!ep || !ep->out) return -1;
  if (!ep->atomic && (N % 8 != 0 || ep->ldc % 8 != 0)) return -1;
  if (ep->atomic && !ep->out_f32) return -1;
  // packed relu' bits exist in the BatchNorm-backward form only, and exclude the tensor form of the same mask
  if (ep->relu_bits && (!(ep->bn_y || ep->mask_after_residual) || ep->dact_aux)) return -1;
  if (ep->residual_subsample < 0 || ep->residual_subsample > 2) return -1;
  if (ep->fp8_out || ep->fp8_scale || ep->fp8_amax) return -1;          // clite_gemm_nt_fp8 only
  return 0;
}

uint32_t span_bytes(int rows, int cols, int ld, size_t es) { return (uint32_t)(((size_t)(rows - 1) * ld + cols) * es); }

// ---- split-K for GEMMs of few output tiles (clite_epilogue.splitk_ws) ------------------------------------------------------------
// The heads' GEMMs have M = 128 or 256 rows: 16-32 output tiles, i.e. 16-32 CUs pulling an 8 MB weight matrix out of HBM at ~25 GB/s each
// (37 us in the step for 128 x 2048 x 2048). Split over K every CU fetches a slice: the partial tiles meet in an f32 workspace through
// float atomics and splitk_finish_kernel applies the epilogue. bf16 only (the exact-f32 mode keeps one k-ordered chain per output).
template <typename T>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ ws, clite_epilogue ep, int M, int N) {
  // workgroup = 8 column chunks (64 columns) x 32 row lanes; all M rows of its columns, so the column statistics need no cross-workgroup step
  __shared__ float red[32][8][16];
  const int tid = threadIdx.x, cl = tid & 7, rl = tid >> 3;
  const int col = (blockIdx.x * 8 + cl) * 8;
  const bool colok = col < N;
  float bias[8], csum[8], csq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { bias[e] = (colok && ep.bias) ? ep.bias[col + e] : 0.f; csum[e] = 0.f; csq[e] = 0.f; }
  if (colok) {
    for (int r = rl; r < M; r += 32) {
      float v[8];
      load8(ws + (size_t)r * N + col, v);
      const size_t o = (size_t)r * ep.ldc + col;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * ep.alpha + bias[e];
      if (ep.preact) store8((T*)ep.preact + o, v);
      if (ep.act == CLITE_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = relu_f(v[e]);
      } else if (ep.act == CLITE_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_t<T>(v[e]);
      } else if (ep.act == CLITE_ACT_TANH) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
      }
      if (ep.dact_aux) {
        float a[8];
        load8((const T*)ep.dact_aux + o, a);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= ep.dact == 1 ? (a[e] > 0.f ? 1.f : 0.f) : ep.dact == 2 ? gelu_grad_t<T>(a[e]) : (1.f - a[e] * a[e]);
      }
      if (ep.residual) {
        float rv[8];
        load8((const T*)ep.residual + o, rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
      }
      if (ep.out_f32) {
        store8((float*)ep.out + o, v);
      } else {
        store8((T*)ep.out + o, v);
        round8_bf16(v);       // statistics of what was stored
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) { csum[e] += v[e]; csq[e] += v[e] * v[e]; }
    }
  }
  if (!ep.colsum) return;
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[rl][cl][e] = csum[e]; red[rl][cl][8 + e] = csq[e]; }
  __syncthreads();
  if (tid < 128) {
    const int c = tid >> 4, e = tid & 15;
    float s = 0.f;
    for (int r = 0; r < 32; ++r) s += red[r][c][e];
    const int cc = (blockIdx.x * 8 + c) * 8 + (e & 7);
    float* crep = ep.colsum + (ep.colsum_replicas > 1 ? (size_t)(blockIdx.x % ep.colsum_replicas) * ep.colsum_stride : 0);
    if (cc < N && (e < 8 || ep.colsum_rows != 1)) atomic_add_f32(crep + (e >= 8 ? N : 0) + cc, s);
  }
}

// number of K splits for the workspace form, or 0 when the launch does not qualify
template <typename T>
int splitk_plan(const clite_epilogue& ep, int M, int N, int ktiles) {
  if (deterministic() || sizeof(T) != 2 || !ep.splitk_ws || ep.atomic || ep.drop_p > 0.f || ep.bn_y || ep.mask_after_residual || N % 8) return 0;
  const int pref = splitk_ws_pref();
  if (!pref) return 0;
  // every split adds a full f32 tile of atomic traffic (64 KB; chip-wide ~1.3 TB/s): ~64 workgroups balance that against the number of
  // CUs fetching the weights (tools/probe_heads.py, sum over the heads' shapes: no split 482 us, 32 -> 371, 64 -> 339, 128 -> 357, 256 -> 425)
  const long target = pref > 1 ? pref : 64;
  long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
  if (tiles > 48 || ktiles < 8) return 0;
  long want = (target + tiles - 1) / tiles, cap = ktiles / 2;
  if (want > cap) want = cap;
  return want >= 2 ? (int)want : 0;
}

// splits of a forward-form product that accumulates into a caller-owned f32 buffer (ep.atomic on gemm_nt / gemm_nn: the loss heads' fused
// MI blocks, loss.py): few output tiles against megabytes of weights, i.e. the workspace form's balance, not the weight gradients'
int fewtile_splits(int M, int N, int ktiles) {
  long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
  if (tiles > 48 || ktiles < 8) return pick_splits(M, N, ktiles);
  const int pref = splitk_ws_pref();
  const long target = pref > 1 ? pref : 64;
  long want = (target + tiles - 1) / tiles, cap = ktiles / 2;
  if (want > cap) want = cap;
  return want >= 1 ? (int)want : 1;
}

template <typename T>
int splitk_finish(const clite_epilogue& ep, int M, int N, hipStream_t st) {
  hipLaunchKernelGGL(splitk_finish_kernel<T>, dim3((N / 8 + 7) / 8), dim3(256), 0, st, (const float*)ep.splitk_ws, ep, M, N);
  return (int)hipGetLastError();
}

clite_epilogue splitk_partial(const clite_epilogue& ep, int N) {
  clite_epilogue e{};
  e.out = ep.splitk_ws; e.ldc = N; e.out_f32 = 1; e.atomic = 1; e.alpha = 1.f;
  return e;
}

template <typename T>
int gemm_nt(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const clite_epilogue* ep, hipStream_t st) {
  constexpr int BK = Cfg<T>::BK;
  uint32_t ab = span_bytes(M, K, lda, sizeof(T)), bb = span_bytes(N, K, ldb, sizeof(T));
  int splits = ep->atomic ? fewtile_splits(M, N, (K + BK - 1) / BK) : 1;
  if (int sk = splitk_plan<T>(*ep, M, N, (K + BK - 1) / BK)) {
    clite_epilogue e1 = splitk_partial(*ep, N);
    GatherKC<T, 128, BK, false> la{A, ab, geom_dense(M, K, lda)};
    GatherKC<T, 128, BK, false> lb{B, bb, geom_dense(N, K, ldb)};
    int rc = launch<T, typename Cfg<T>::C128>(la, lb, e1, M, N, K, sk, st);
    return rc ? rc : splitk_finish<T>(*ep, M, N, st);
  }
  if (N <= 64) {
    GatherKC<T, 256, BK, false> la{A, ab, geom_dense(M, K, lda)};
    GatherKC<T, 64, BK, false> lb{B, bb, geom_dense(N, K, ldb)};
    return launch<T, typename Cfg<T>::C256x64>(la, lb, *ep, M, N, K, splits, st);
  }
  GatherKC<T, 128, BK, false> la{A, ab, geom_dense(M, K, lda)};
  GatherKC<T, 128, BK, false> lb{B, bb, geom_dense(N, K, ldb)};
  return launch<T, typename Cfg<T>::C128>(la, lb, *ep, M, N, K, splits, st);
}

template <typename T>
int gemm_nn(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const clite_epilogue* ep, hipStream_t st) {
  constexpr int BK = Cfg<T>::BK;
  uint32_t ab = span_bytes(M, K, lda, sizeof(T)), bb = span_bytes(K, N, ldb, sizeof(T));
  int splits = ep->atomic ? fewtile_splits(M, N, (K + BK - 1) / BK) : 1;
  if (int sk = splitk_plan<T>(*ep, M, N, (K + BK - 1) / BK)) {
    clite_epilogue e1 = splitk_partial(*ep, N);
    GatherKC<T, 128, BK, false> la{A, ab, geom_dense(M, K, lda)};
    StridedXC<T, 128, BK> lb{B, bb, ldb, N, K, 1};
    int rc = launch<T, typename Cfg<T>::C128>(la, lb, e1, M, N, K, sk, st);
    return rc ? rc : splitk_finish<T>(*ep, M, N, st);
  }
  if (N <= 64) {
    GatherKC<T, 256, BK, false> la{A, ab, geom_dense(M, K, lda)};
    StridedXC<T, 64, BK> lb{B, bb, ldb, N, K, 1};
    return launch<T, typename Cfg<T>::C256x64>(la, lb, *ep, M, N, K, splits, st);
  }
  GatherKC<T, 128, BK, false> la{A, ab, geom_dense(M, K, lda)};
  StridedXC<T, 128, BK> lb{B, bb, ldb, N, K, 1};
  return launch<T, typename Cfg<T>::C128>(la, lb, *ep, M, N, K, splits, st);
}

template <typename T>
int gemm_tn(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const clite_epilogue* ep, hipStream_t st) {
  constexpr int BK = Cfg<T>::BK;
  uint32_t ab = span_bytes(K, M, lda, sizeof(T)), bb = span_bytes(K, N, ldb, sizeof(T));
  int splits = ep->atomic ? pick_splits(M, N, (K + BK - 1) / BK) : 1;
  StridedXC<T, 128, BK> la{A, ab, lda, M, K, 1};
  StridedXC<T, 128, BK> lb{B, bb, ldb, N, K, 1};
  return launch<T, typename Cfg<T>::C128>(la, lb, *ep, M, N, K, splits, st);
}

int check_conv(const clite_conv* c) {
  if (!c) return -1;
  if (c->dtype != CLITE_BF16 && c->dtype != CLITE_F32) return -1;
  if (c->C % 8 || c->K % 8) return -1;
  if (c->R * c->S > 1 && (c->C % 32 || c->K % 32)) return -1;   // a K tile must stay inside one (r,s)
  if (c->Ho != (c->H + 2 * c->pad - c->R) / c->stride + 1) return -1;
  if (c->Wo != (c->W + 2 * c->pad - c->S) / c->stride + 1) return -1;
  if (!fits32((size_t)c->N * c->H * c->W * c->C, 4) || !fits32((size_t)c->N * c->Ho * c->Wo * c->K, 4)) return -1;
  return 0;
}
ConvGeom geom_fwd(const clite_conv& c) {   // rows = output pixels, gather x
  ConvGeom g;
  g.H = c.H; g.W = c.W; g.C = c.C;
  g.sN = c.H * c.W * c.C; g.sH = c.W * c.C; g.sW = c.C;
  g.RH = c.Ho; g.RW = c.Wo; g.R = c.R; g.S = c.S; g.stride = c.stride; g.pad = c.pad; g.padw = c.pad;
  g.rows = c.N * c.Ho * c.Wo; g.concat = 0;
  g.div_hw = fastdiv_make(c.Ho * c.Wo);
  g.div_w = fastdiv_make(c.Wo);
  return g;
}
ConvGeom geom_dgrad(const clite_conv& c) {  // rows = input pixels, gather dy
  ConvGeom g;
  g.H = c.Ho; g.W = c.Wo; g.C = c.K;
  g.sN = c.Ho * c.Wo * c.K; g.sH = c.Wo * c.K; g.sW = c.K;
  g.RH = c.H; g.RW = c.W; g.R = c.R; g.S = c.S; g.stride = c.stride; g.pad = c.pad; g.padw = c.pad;
  g.rows = c.N * c.H * c.W; g.concat = 0;
  g.div_hw = fastdiv_make(c.H * c.W);
  g.div_w = fastdiv_make(c.W);
  return g;
}

template <typename T>
int conv_fwd(const void* x, const void* w, const clite_conv& c, const clite_epilogue* ep, hipStream_t st) {
  constexpr int BK = Cfg<T>::BK;
  int M = c.N * c.Ho * c.Wo, Ktot = c.R * c.S * c.C;
  uint32_t xb = (uint32_t)((size_t)c.N * c.H * c.W * c.C * sizeof(T)), wb = (uint32_t)((size_t)c.K * Ktot * sizeof(T));
  if constexpr (sizeof(T) == 2) {
    // 3 x 3 / stride 1, 64 -> 64: the patch-resident kernel (conv_patch.hip) — automatic policy only; the deterministic mode keeps the path whose
    // statistics come from colstats_det
    if (use_dma() && !deterministic() && tile_policy_value() == 0) {
      const int rc = launch_conv3x3_patch(x, w, c, *ep, false, st);
      if (rc != WIDE_NOT_TAKEN) return rc;
    }
  }
  if (c.K <= 64) {
    GatherKC<T, 256, BK, false> la{x, xb, geom_fwd(c)};
    GatherKC<T, 64, BK, false> lb{w, wb, geom_dense(c.K, Ktot)};
    return launch<T, typename Cfg<T>::C256x64>(la, lb, *ep, M, c.K, Ktot, 1, st);
  }
  GatherKC<T, 128, BK, false> la{x, xb, geom_fwd(c)};
  GatherKC<T, 128, BK, false> lb{w, wb, geom_dense(c.K, Ktot)};
  return launch<T, typename Cfg<T>::C128>(la, lb, *ep, M, c.K, Ktot, 1, st);
}

// WT: the weight is given transposed, [C][R][S][K] — a k-contiguous (KC) operand like the forward's, instead of the k-strided XC image
template <typename T, bool WT = false>
int conv_dgrad(const void* dy, const void* w, const clite_conv& c, const clite_epilogue* ep, hipStream_t st) {
  constexpr int BK = Cfg<T>::BK;
  int M = c.N * c.H * c.W, Ktot = c.R * c.S * c.K;
  uint32_t yb = (uint32_t)((size_t)c.N * c.Ho * c.Wo * c.K * sizeof(T)), wb = (uint32_t)((size_t)c.K * c.R * c.S * c.C * sizeof(T));
  if constexpr (WT) {
    if constexpr (sizeof(T) == 2) {
      if (use_dma() && !deterministic() && tile_policy_value() == 0) {          // 3 x 3 / stride 1, 64 <- 64: conv_patch.hip
        const int rc = launch_conv3x3_patch(dy, w, c, *ep, true, st);
        if (rc != WIDE_NOT_TAKEN) return rc;
      }
    }
    if (c.R == 1 && c.S == 1 && c.pad == 0 && c.stride > 1 && ep->residual == ep->out && !ep->colsum && !ep->preact && !ep->dact_aux) {
      int P = c.N * c.Ho * c.Wo;      // strided 1x1 shortcut: dense GEMM over the output pixels, rows scattered (see below)
      RowMap rm;
      rm.on = 1; rm.div_hw = fastdiv_make(c.Ho * c.Wo); rm.div_w = fastdiv_make(c.Wo); rm.H = c.H; rm.W = c.W; rm.stride = c.stride; rm.off_h = 0; rm.off_w = 0;
      if (c.C <= 64) {
        GatherKC<T, 256, BK, false> la{dy, yb, geom_dense(P, c.K)};
        GatherKC<T, 64, BK, false> lb{w, wb, geom_dense(c.C, c.K)};
        return launch<T, typename Cfg<T>::C256x64>(la, lb, *ep, P, c.C, c.K, 1, st, rm);
      }
      GatherKC<T, 128, BK, false> la{dy, yb, geom_dense(P, c.K)};
      GatherKC<T, 128, BK, false> lb{w, wb, geom_dense(c.C, c.K)};
      return launch<T, typename Cfg<T>::C128>(la, lb, *ep, P, c.C, c.K, 1, st, rm);
    }
    RowMap rm{};
    if (ep->residual_subsample > 1) {
      // clite_epilogue.residual_subsample: the compact gradient of a stride-2 shortcut as the residual of this (dense, 1 x 1 / stride 1) dgrad
      if (ep->residual_subsample != 2 || c.R != 1 || c.S != 1 || c.stride != 1 || c.pad != 0 || (c.H & 1) || (c.W & 1) || !ep->residual || !ep->bn_y ||
          !ep->mask_after_residual)
        return -1;
      rm.on = 2; rm.div_hw = fastdiv_make(c.H * c.W); rm.div_w = fastdiv_make(c.W); rm.H = c.H; rm.W = c.W; rm.stride = 2; rm.off_h = 0; rm.off_w = 0;
    }
    if (c.C <= 64) {
      GatherKC<T, 256, BK, true> la{dy, yb, geom_dgrad(c)};
      GatherKC<T, 64, BK, false> lb{w, wb, geom_dense(c.C, Ktot)};
      return launch<T, typename Cfg<T>::C256x64>(la, lb, *ep, M, c.C, Ktot, 1, st, rm);
    }
    GatherKC<T, 128, BK, true> la{dy, yb, geom_dgrad(c)};
    GatherKC<T, 128, BK, false> lb{w, wb, geom_dense(c.C, Ktot)};
    return launch<T, typename Cfg<T>::C128>(la, lb, *ep, M, c.C, Ktot, 1, st, rm);
  }
  if (ep->residual_subsample > 1) return -1;          // (the transposed-weight entry point only)
  if (c.R == 1 && c.S == 1 && c.pad == 0 && c.stride > 1 && ep->residual == ep->out && !ep->colsum && !ep->preact && !ep->dact_aux) {
    // 1x1 / stride-s shortcut conv accumulated in place (dx += dgrad): only every s-th pixel of dx receives a contribution, so run
    // the dense GEMM dy[P][K] * W[K][C] over the P output pixels and scatter-add its rows (RowMap) instead of gathering a mostly
    // empty im2col over all N*H*W input pixels (s*s times the work)
    int P = c.N * c.Ho * c.Wo;
    RowMap rm;
    rm.on = 1; rm.div_hw = fastdiv_make(c.Ho * c.Wo); rm.div_w = fastdiv_make(c.Wo); rm.H = c.H; rm.W = c.W; rm.stride = c.stride; rm.off_h = 0; rm.off_w = 0;
    if (c.C <= 64) {
      GatherKC<T, 256, BK, false> la{dy, yb, geom_dense(P, c.K)};
      StridedXC<T, 64, BK> lb{w, wb, c.C, c.C, c.K, 1};
      return launch<T, typename Cfg<T>::C256x64>(la, lb, *ep, P, c.C, c.K, 1, st, rm);
    }
    GatherKC<T, 128, BK, false> la{dy, yb, geom_dense(P, c.K)};
    StridedXC<T, 128, BK> lb{w, wb, c.C, c.C, c.K, 1};
    return launch<T, typename Cfg<T>::C128>(la, lb, *ep, P, c.C, c.K, 1, st, rm);
  }
  if (c.C <= 64) {
    GatherKC<T, 256, BK, true> la{dy, yb, geom_dgrad(c)};
    StridedXC<T, 64, BK> lb{w, wb, c.R * c.S * c.C, c.C, c.K, c.R * c.S};
    return launch<T, typename Cfg<T>::C256x64>(la, lb, *ep, M, c.C, Ktot, 1, st);
  }
  GatherKC<T, 128, BK, true> la{dy, yb, geom_dgrad(c)};
  StridedXC<T, 128, BK> lb{w, wb, c.R * c.S * c.C, c.C, c.K, c.R * c.S};
  return launch<T, typename Cfg<T>::C128>(la, lb, *ep, M, c.C, Ktot, 1, st);
}

// One parity class (ph, pw) of the dgrad of a 3x3 / stride-2 / pad-1 conv: input pixels (2*hq + ph, 2*wq + pw) only receive the
// taps r = r0 + 2a, s = s0 + 2b with r0 = (ph + 1) & 1, s0 = (pw + 1) & 1, and ho = hq + ch - a with ch = (ph + 1 - r0) / 2: a
// stride-1 dgrad with an na x nb window and padding (ch, cw) over the [N][H/2][W/2] sub-grid, scattered to its pixels by RowMap.
// `wsub` holds that class's taps packed as [K][na][nb][C]. The four classes together cost 9 taps per output quad instead of the
// 36 that a gather over all pixels and all taps (3/4 of them structurally zero) performs.
template <typename T, bool WT = false>
int conv_dgrad_s2class(const void* dy, const void* wsub, const clite_conv& c, int ph, int pw, const clite_epilogue* ep, hipStream_t st) {
  constexpr int BK = Cfg<T>::BK;
  const int r0 = (ph + 1) & 1, s0 = (pw + 1) & 1;
  const int na = r0 ? 1 : 2, nb = s0 ? 1 : 2;
  const int ch = (ph + 1 - r0) / 2, cw = (pw + 1 - s0) / 2;
  const int Hq = c.H / 2, Wq = c.W / 2;
  ConvGeom g;
  g.H = c.Ho; g.W = c.Wo; g.C = c.K;
  g.sN = c.Ho * c.Wo * c.K; g.sH = c.Wo * c.K; g.sW = c.K;
  g.RH = Hq; g.RW = Wq; g.R = na; g.S = nb; g.stride = 1; g.pad = ch; g.padw = cw;
  g.rows = c.N * Hq * Wq; g.concat = 0;
  g.div_hw = fastdiv_make(Hq * Wq);
  g.div_w = fastdiv_make(Wq);
  RowMap rm;
  rm.on = 1; rm.div_hw = g.div_hw; rm.div_w = g.div_w; rm.H = c.H; rm.W = c.W; rm.stride = 2; rm.off_h = ph; rm.off_w = pw;
  const int M = g.rows, Ktot = na * nb * c.K;
  uint32_t yb = (uint32_t)((size_t)c.N * c.Ho * c.Wo * c.K * sizeof(T)), wb = (uint32_t)((size_t)c.K * na * nb * c.C * sizeof(T));
  if constexpr (WT) {            // wsub = [C][na][nb][K]
    if (c.C <= 64) {
      GatherKC<T, 256, BK, true> la{dy, yb, g};
      GatherKC<T, 64, BK, false> lb{wsub, wb, geom_dense(c.C, Ktot)};
      return launch<T, typename Cfg<T>::C256x64>(la, lb, *ep, M, c.C, Ktot, 1, st, rm);
    }
    GatherKC<T, 128, BK, true> la{dy, yb, g};
    GatherKC<T, 128, BK, false> lb{wsub, wb, geom_dense(c.C, Ktot)};
    return launch<T, typename Cfg<T>::C128>(la, lb, *ep, M, c.C, Ktot, 1, st, rm);
  }
  if (c.C <= 64) {
    GatherKC<T, 256, BK, true> la{dy, yb, g};
    StridedXC<T, 64, BK> lb{wsub, wb, na * nb * c.C, c.C, c.K, na * nb};
    return launch<T, typename Cfg<T>::C256x64>(la, lb, *ep, M, c.C, Ktot, 1, st, rm);
  }
  GatherKC<T, 128, BK, true> la{dy, yb, g};
  StridedXC<T, 128, BK> lb{wsub, wb, na * nb * c.C, c.C, c.K, na * nb};
  return launch<T, typename Cfg<T>::C128>(la, lb, *ep, M, c.C, Ktot, 1, st, rm);
}

template <typename T>
int conv_wgrad(const void* dy, const void* x, const clite_conv& c, float* dw, hipStream_t st) {
  constexpr int BK = Cfg<T>::BK;
  int P = c.N * c.Ho * c.Wo, Ncols = c.R * c.S * c.C;
  uint32_t yb = (uint32_t)((size_t)P * c.K * sizeof(T)), xb = (uint32_t)((size_t)c.N * c.H * c.W * c.C * sizeof(T));
  clite_epilogue ep = {};
  ep.out = dw; ep.ldc = Ncols; ep.out_f32 = 1; ep.atomic = 1; ep.alpha = 1.f;
  if (c.K <= 64) {          // a 128-row tile would be half empty
    StridedXC<T, 64, BK> la{dy, yb, c.K, c.K, P, 1};
    GatherXC<T, 128, BK> lb{x, xb, geom_fwd(c)};
    return launch<T, typename Cfg<T>::C64x128>(la, lb, ep, c.K, Ncols, P, pick_splits(c.K, Ncols, (P + BK - 1) / BK, 64, 128, c.R * c.S > 1), st);
  }
  if (Ncols <= 64) {
    StridedXC<T, 128, BK> la{dy, yb, c.K, c.K, P, 1};
    GatherXC<T, 64, BK> lb{x, xb, geom_fwd(c)};
    return launch<T, typename Cfg<T>::C128x64>(la, lb, ep, c.K, Ncols, P, pick_splits(c.K, Ncols, (P + BK - 1) / BK, 128, 64, c.R * c.S > 1), st);
  }
  StridedXC<T, 128, BK> la{dy, yb, c.K, c.K, P, 1};
  GatherXC<T, 128, BK> lb{x, xb, geom_fwd(c)};
  int splits = pick_splits(c.K, Ncols, (P + BK - 1) / BK, 128, 128, c.R * c.S > 1);
  return launch<T, typename Cfg<T>::C128>(la, lb, ep, c.K, Ncols, P, splits, st);
}

// 7x7/2 stem on a pre-padded NHWC4 image [N][Hp][Wp][4]: a 7x1 window over 32 "virtual" channels (8 adjacent pixels
// x 4 channels are contiguous), K = 7*32 = 224; weights packed as [64][7][8][4] with s=7 and c=3 zero.
ConvGeom geom_stem(int N, int Hp, int Wp, int Ho, int Wo) {
  ConvGeom g;
  g.H = Hp; g.W = Wp; g.C = 32;
  g.sN = Hp * Wp * 4; g.sH = Wp * 4; g.sW = 4;
  g.RH = Ho; g.RW = Wo; g.R = 7; g.S = 1; g.stride = 2; g.pad = 0; g.padw = 0;
  g.rows = N * Ho * Wo; g.concat = 0;
  g.div_hw = fastdiv_make(Ho * Wo);
  g.div_w = fastdiv_make(Wo);
  return g;
}
int check_stem(int dtype, int N, int Hp, int Wp, int Ho, int Wo) {
  if (dtype != CLITE_BF16 && dtype != CLITE_F32) return -1;
  if (N <= 0 || Wp % 2 || 2 * (Ho - 1) + 7 > Hp || 2 * (Wo - 1) + 8 > Wp) return -1;
  if (!fits32((size_t)N * Hp * Wp * 4, 4) || !fits32((size_t)N * Ho * Wo * 64, 4)) return -1;
  return 0;
}
template <typename T>
int stem_fwd(const void* xpad, const void* wv, int N, int Hp, int Wp, int Ho, int Wo, const clite_epilogue* ep, hipStream_t st) {
  constexpr int BK = Cfg<T>::BK;
  GatherKC<T, 256, BK, false> la{xpad, (uint32_t)((size_t)N * Hp * Wp * 4 * sizeof(T)), geom_stem(N, Hp, Wp, Ho, Wo)};
  GatherKC<T, 64, BK, false> lb{wv, (uint32_t)(64 * 224 * sizeof(T)), geom_dense(64, 224)};
  return launch<T, typename Cfg<T>::C256x64>(la, lb, *ep, N * Ho * Wo, 64, 224, 1, st);
}
template <typename T>
int stem_wgrad(const void* dy, const void* xpad, int N, int Hp, int Wp, int Ho, int Wo, float* dwv, hipStream_t st) {
  constexpr int BK = Cfg<T>::BK;
  int P = N * Ho * Wo;
  clite_epilogue ep = {};
  ep.out = dwv; ep.ldc = 224; ep.out_f32 = 1; ep.atomic = 1; ep.alpha = 1.f;
  // 64 output channels: the 64-row tile of conv_wgrad (a 128-row tile is half empty: twice the MFMAs and dy fragment reads; 198.7 -> 181.2 us)
  StridedXC<T, 64, BK> la{dy, (uint32_t)((size_t)P * 64 * sizeof(T)), 64, 64, P, 1};
  GatherXC<T, 128, BK> lb{xpad, (uint32_t)((size_t)N * Hp * Wp * 4 * sizeof(T)), geom_stem(N, Hp, Wp, Ho, Wo)};
  int splits = pick_splits(64, 224, (P + BK - 1) / BK, 64, 128, true);
  return launch<T, typename Cfg<T>::C64x128>(la, lb, ep, 64, 224, P, splits, st);
}

// w f32 [64][7][7][3] (KRSC) -> wv T [64][7][8][4]; dwv f32 [64][7][8][4] -> dw f32 [64][7][7][3] (+=)
template <typename T>
__global__ __launch_bounds__(256) void stem_pack_kernel(const float* w, T* wv) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 64 * 7 * 8 * 4) return;
  int c = i & 3, s = (i >> 2) & 7, r = (i >> 5) % 7, k = i / 224;
  float v = (c < 3 && s < 7) ? w[((k * 7 + r) * 7 + s) * 3 + c] : 0.f;
  if constexpr (sizeof(T) == 2) wv[i] = f2bf(v); else wv[i] = v;
}
__global__ __launch_bounds__(256) void stem_unpack_kernel(const float* dwv, float* dw) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 64 * 7 * 7 * 3) return;
  int c = i % 3, s = (i / 3) % 7, r = (i / 21) % 7, k = i / 147;
  dw[i] += dwv[((k * 7 + r) * 8 + s) * 4 + c];
}

int check_gemm(const clite_epilogue* ep, int dtype, int M, int N, int K, int lda, int ldb, int rows_a, int rows_b) {
  if (ep && ep->residual_subsample > 1) return -1;          // clite_conv_dgrad_wt only
  if (dtype != CLITE_BF16 && dtype != CLITE_F32) return -1;
  if (M <= 0 || N <= 0 || K <= 0 || lda % 8 || ldb % 8) return -1;
  if (!fits32((size_t)rows_a * lda, 4) || !fits32((size_t)rows_b * ldb, 4)) return -1;
  return check_ep(ep, N);
}

}  // namespace

extern "C" int clite_abi_version(void) { return CLITE_ABI_VERSION; }
extern "C" int clite_set_deterministic(int on) { g_deterministic.store(on ? 1 : 0, std::memory_order_relaxed); return 0; }
extern "C" int clite_get_deterministic(void) { return deterministic() ? 1 : 0; }
extern "C" int clite_set_f32_split(int on) { g_f32_split.store(on ? 1 : 0, std::memory_order_relaxed); return 0; }
extern "C" int clite_get_f32_split(void) { return f32_split() ? 1 : 0; }
#if CLITE_STAMP
extern "C" int clite_dbg_read(unsigned long long* host, int n) {     // diagnostic builds only
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(clite_dbg), (size_t)n * 8, 0, hipMemcpyDeviceToHost);
}
#endif

extern "C" int clite_gemm_nt(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int dtype, const clite_epilogue* ep, void* stream) {
  if (check_gemm(ep, dtype, M, N, K, lda, ldb, M, N) || K % 8 || lda < K || ldb < K) return -1;
  return dtype == CLITE_BF16 ? gemm_nt<bf16>(A, lda, B, ldb, M, N, K, ep, (hipStream_t)stream) : gemm_nt<float>(A, lda, B, ldb, M, N, K, ep, (hipStream_t)stream);
}
extern "C" int clite_gemm_nn(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int dtype, const clite_epilogue* ep, void* stream) {
  if (check_gemm(ep, dtype, M, N, K, lda, ldb, M, K) || K % 8 || lda < K || ldb < N) return -1;
  return dtype == CLITE_BF16 ? gemm_nn<bf16>(A, lda, B, ldb, M, N, K, ep, (hipStream_t)stream) : gemm_nn<float>(A, lda, B, ldb, M, N, K, ep, (hipStream_t)stream);
}
extern "C" int clite_gemm_tn(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int dtype, const clite_epilogue* ep, void* stream) {
  if (check_gemm(ep, dtype, M, N, K, lda, ldb, K, K) || M % 8 || N % 8 || lda < M || ldb < N) return -1;
  return dtype == CLITE_BF16 ? gemm_tn<bf16>(A, lda, B, ldb, M, N, K, ep, (hipStream_t)stream) : gemm_tn<float>(A, lda, B, ldb, M, N, K, ep, (hipStream_t)stream);
}
extern "C" int clite_conv_fwd(const void* x, const void* w, const clite_conv* cv, const clite_epilogue* ep, void* stream) {
  if (check_conv(cv) || check_ep(ep, cv->K) || ep->residual_subsample > 1) return -1;
  return cv->dtype == CLITE_BF16 ? conv_fwd<bf16>(x, w, *cv, ep, (hipStream_t)stream) : conv_fwd<float>(x, w, *cv, ep, (hipStream_t)stream);
}
extern "C" int clite_conv_dgrad(const void* dy, const void* w, const clite_conv* cv, const clite_epilogue* ep, void* stream) {
  if (check_conv(cv) || check_ep(ep, cv->C)) return -1;
  return cv->dtype == CLITE_BF16 ? conv_dgrad<bf16>(dy, w, *cv, ep, (hipStream_t)stream) : conv_dgrad<float>(dy, w, *cv, ep, (hipStream_t)stream);
}
extern "C" int clite_conv_dgrad_s2class(const void* dy, const void* wsub, const clite_conv* cv, int ph, int pw, const clite_epilogue* ep, void* stream) {
  if (check_conv(cv) || check_ep(ep, cv->C)) return -1;
  if (cv->R != 3 || cv->S != 3 || cv->stride != 2 || cv->pad != 1 || (cv->H & 1) || (cv->W & 1) || (unsigned)ph > 1 || (unsigned)pw > 1) return -1;
  if (cv->C % 64 || cv->K % 64 || ep->residual_subsample > 1) return -1;
  return cv->dtype == CLITE_BF16 ? conv_dgrad_s2class<bf16>(dy, wsub, *cv, ph, pw, ep, (hipStream_t)stream)
                                 : conv_dgrad_s2class<float>(dy, wsub, *cv, ph, pw, ep, (hipStream_t)stream);
}
extern "C" int clite_conv_dgrad_wt(const void* dy, const void* wt, const clite_conv* cv, const clite_epilogue* ep, void* stream) {
  if (check_conv(cv) || check_ep(ep, cv->C)) return -1;
  return cv->dtype == CLITE_BF16 ? conv_dgrad<bf16, true>(dy, wt, *cv, ep, (hipStream_t)stream) : conv_dgrad<float, true>(dy, wt, *cv, ep, (hipStream_t)stream);
}
// The block-output BatchNorm backward folded into the 1 x 1 convolution's input gradient (include/clite.h, ABI v12): A = [dz | y] as a two-slot "window" over
// the pair buffer [2][M][K] — the dgrad gather with R = 2, S = 1, pad 1 over a 2 x M "image" whose single output row is the M pixels: r = 0 reads slot 1 (y),
// r = 1 slot 0 (dz) — B = w2 [Cin][2][K] (k-contiguous rows of 2K), the BatchNorm-backward epilogue with the constant row as its bias.
extern "C" int clite_conv_dgrad_bnfold(const void* pair, const void* w2, int M, int K, int Cin, const clite_epilogue* ep, void* stream) {
  typedef bf16 T;
  constexpr int BK = Cfg<T>::BK;
  if (!pair || !w2 || M <= 0 || K <= 0 || Cin <= 0 || K % 64 || Cin % 8 || check_ep(ep, Cin)) return -1;
  if (!ep->bn_y || !ep->relu_bits || !ep->bias || ep->residual || ep->mask_after_residual || ep->out_f32 || ep->alpha != 1.f || ep->residual_subsample > 1) return -1;
  if (!fits32((size_t)2 * M * K, sizeof(T)) || (size_t)2 * M * K >= ((size_t)1 << 31)) return -1;
  ConvGeom g;
  g.H = 2; g.W = M; g.C = K;
  g.sN = 0; g.sH = M * K; g.sW = K;
  g.RH = 1; g.RW = M; g.R = 2; g.S = 1; g.stride = 1; g.pad = 1; g.padw = 0;
  g.rows = M; g.concat = 1;
  g.div_hw = fastdiv_make(M);
  g.div_w = fastdiv_make(M);
  const uint32_t ab = (uint32_t)((size_t)2 * M * K * sizeof(T)), wb = (uint32_t)((size_t)Cin * 2 * K * sizeof(T));
  hipStream_t st = (hipStream_t)stream;
  if (use_dma()) {          // K = 256, Cin = 64 (the 56 x 56 stage): the streaming kernel of fold_dgrad.hip
    const int rc = launch_fold_dgrad_rows(pair, w2, M, K, Cin, *ep, st);
    if (rc != WIDE_NOT_TAKEN) return rc;
  }
  if (Cin <= 64) {
    GatherKC<T, 256, BK, true> la{pair, ab, g};
    GatherKC<T, 64, BK, false> lb{w2, wb, geom_dense(Cin, 2 * K)};
    return launch<T, typename Cfg<T>::C256x64>(la, lb, *ep, M, Cin, 2 * K, 1, st);
  }
  GatherKC<T, 128, BK, true> la{pair, ab, g};
  GatherKC<T, 128, BK, false> lb{w2, wb, geom_dense(Cin, 2 * K)};
  return launch<T, typename Cfg<T>::C128>(la, lb, *ep, M, Cin, 2 * K, 1, st);
}
extern "C" int clite_conv_dgrad_s2class_wt(const void* dy, const void* wtsub, const clite_conv* cv, int ph, int pw, const clite_epilogue* ep, void* stream) {
  if (check_conv(cv) || check_ep(ep, cv->C)) return -1;
  if (cv->R != 3 || cv->S != 3 || cv->stride != 2 || cv->pad != 1 || (cv->H & 1) || (cv->W & 1) || (unsigned)ph > 1 || (unsigned)pw > 1) return -1;
  if (cv->C % 64 || cv->K % 64 || ep->residual_subsample > 1) return -1;
  return cv->dtype == CLITE_BF16 ? conv_dgrad_s2class<bf16, true>(dy, wtsub, *cv, ph, pw, ep, (hipStream_t)stream)
                                 : conv_dgrad_s2class<float, true>(dy, wtsub, *cv, ph, pw, ep, (hipStream_t)stream);
}
extern "C" int clite_conv_wgrad(const void* dy, const void* x, const clite_conv* cv, float* dw, void* stream) {
  if (check_conv(cv) || !dw) return -1;
  return cv->dtype == CLITE_BF16 ? conv_wgrad<bf16>(dy, x, *cv, dw, (hipStream_t)stream) : conv_wgrad<float>(dy, x, *cv, dw, (hipStream_t)stream);
}

// The 64 -> 64 3 x 3 / stride 1 weight gradient on the patch-resident kernel (conv_patch.hip). Returns 1 when the problem is not one it covers
// (shape, dtype, workspace too small, deterministic mode, a forced tile policy): the caller then takes clite_conv_wgrad / the grouped launch.
extern "C" int clite_conv_wgrad_patch_workspace(uint64_t* nbytes) {
  if (!nbytes) return -1;
  *nbytes = (uint64_t)conv3x3_wgrad_patch_workspace();
  return 0;
}
extern "C" int clite_conv_wgrad_patch(const void* dy, const void* x, const clite_conv* cv, float* dw, void* ws, uint64_t ws_bytes, void* stream) {
  if (check_conv(cv) || !dw) return -1;
  if (deterministic() || tile_policy_value() != 0) return 1;
  const int rc = launch_conv3x3_wgrad_patch(dy, x, *cv, dw, ws, (size_t)ws_bytes, (hipStream_t)stream);
  return rc == WIDE_NOT_TAKEN ? 1 : rc;
}

// The stem's weight gradient on the patch-resident kernel (conv_patch.hip), straight into the [64][7][7][3] f32 gradient (+=). Returns 1 when it does
// not cover the problem (f32, shape, workspace, deterministic mode, a forced tile policy): the caller then takes clite_stem_wgrad + clite_stem_unpack_grad.
extern "C" int clite_stem_wgrad_patch(const void* dy, const void* xpad, int dtype, int N, int Hp, int Wp, int Ho, int Wo, float* dw, void* ws, uint64_t ws_bytes,
                                      void* stream) {
  if (check_stem(dtype, N, Hp, Wp, Ho, Wo) || !dy || !xpad || !dw) return -1;
  if (dtype != CLITE_BF16 || deterministic() || tile_policy_value() != 0) return 1;
  const int rc = launch_stem_wgrad_patch(dy, xpad, N, Hp, Wp, Ho, Wo, dw, ws, (size_t)ws_bytes, (hipStream_t)stream);
  return rc == WIDE_NOT_TAKEN ? 1 : rc;
}

extern "C" int clite_stem_fwd(const void* xpad, const void* wv, int dtype, int N, int Hp, int Wp, int Ho, int Wo, const clite_epilogue* ep, void* stream) {
  if (check_stem(dtype, N, Hp, Wp, Ho, Wo) || check_ep(ep, 64)) return -1;
  return dtype == CLITE_BF16 ? stem_fwd<bf16>(xpad, wv, N, Hp, Wp, Ho, Wo, ep, (hipStream_t)stream)
                             : stem_fwd<float>(xpad, wv, N, Hp, Wp, Ho, Wo, ep, (hipStream_t)stream);
}
extern "C" int clite_stem_wgrad(const void* dy, const void* xpad, int dtype, int N, int Hp, int Wp, int Ho, int Wo, float* dwv, void* stream) {
  if (check_stem(dtype, N, Hp, Wp, Ho, Wo) || !dwv) return -1;
  return dtype == CLITE_BF16 ? stem_wgrad<bf16>(dy, xpad, N, Hp, Wp, Ho, Wo, dwv, (hipStream_t)stream)
                             : stem_wgrad<float>(dy, xpad, N, Hp, Wp, Ho, Wo, dwv, (hipStream_t)stream);
}
extern "C" int clite_stem_pack(const float* w, void* wv, int dtype, void* stream) {
  if (!w || !wv) return -1;
  if (dtype == CLITE_BF16) hipLaunchKernelGGL(stem_pack_kernel<bf16>, dim3(56), dim3(256), 0, (hipStream_t)stream, w, (bf16*)wv);
  else if (dtype == CLITE_F32) hipLaunchKernelGGL(stem_pack_kernel<float>, dim3(56), dim3(256), 0, (hipStream_t)stream, w, (float*)wv);
  else return -1;
  return (int)hipGetLastError();
}
extern "C" int clite_stem_unpack_grad(const float* dwv, float* dw, void* stream) {
  if (!dwv || !dw) return -1;
  hipLaunchKernelGGL(stem_unpack_kernel, dim3(37), dim3(256), 0, (hipStream_t)stream, dwv, dw);
  return (int)hipGetLastError();
}
