// Wide-K, 8-wave form of the implicit-GEMM engine (bf16 only; same operands, epilogue semantics and C ABI as igemm_dma.h).
//
// What round 1's kernels (128 x 128 x 32 tiles, 4 waves, 3 workgroups per CU) left on the table, measured with tools/micro/tile_loop.hip
// (the main loop's memory side alone, profiles/r2_tile_loop.txt):
//   * a K tile of 32 bf16 is a 64-BYTE row: every L2 line (128 B) is requested twice, half a line at a time. With 128-byte rows
//     (K tile = 64) the same 128 x 128 tile stages 15.4 TB/s instead of 10.7 (BERT FFN2 shape: 26.6 -> 18.4 us of memory side);
//   * 256-row tiles stage 25-50 % fewer bytes per MFMA (256 x 128: FFN1 22.5 -> 14.9 us; 256 x 256: 13.3 us).
// A 3-stage ring of 128-byte rows is 96-144 KB, i.e. ONE workgroup per CU, so the workgroup itself must keep two waves on every SIMD:
// 512 threads = 8 waves, arranged per tile shape as
//   W128 : 128 x 128, 2 x 2 waves of 64 x 64, x 2 K-GROUPS: both groups work on the same ring stage, group g on k-steps {2g, 2g+1} of the
//          64-deep tile (each wave reads only its half of the K depth from LDS); group 1 hands its accumulators to group 0 at the end
//   W256x128 : 256 x 128, 4 x 2 waves of 64 x 64
//   W256     : 256 x 256, 2 x 4 waves of 128 x 64 (2-stage ring: 128 KB)
// Operand images:
//   KC (k contiguous: activations forward, dY in dgrad, weights forward)   [rows][128 B], 16-byte chunk c of row r in slot c ^ ((r>>1)&7):
//       conflict-free ds_read_b128 fragments (a 16-lane group covers 8 even + 8 odd rows -> 16 distinct slots of the 256-byte bank row)
//   XC (contraction index slow: both operands of wgrad, weights in dgrad)  [64 k][cols * 2 B] — igemm_dma.h's image, 64 k-rows deep,
//       fragments by ds_read_b64_tr_b16.
#ifndef CLITE_IGEMM_WIDE_H
#define CLITE_IGEMM_WIDE_H
#include "igemm_dma.h"

namespace clite {

constexpr int WIDE_NW = 8;       // waves per workgroup
constexpr int WIDE_BK = 64;      // K tile (bf16 elements) = one 128-byte line per row

// ---- KC gather, 128-byte rows. One wave instruction = 8 rows x 128 B.
template <int ROWS, bool DGRAD>
struct WideKC {
  typedef bf16 T;
  static constexpr int EPC = 8;
  static constexpr int NI = ROWS / (8 * WIDE_NW);      // DMA instructions per wave per tile
  static constexpr int BYTES = ROWS * 128;
  static_assert(NI >= 1, "tile too small for 8 waves");
  const void* ptr;
  uint32_t bytes;
  ConvGeom g;

  struct State {
    rsrc_t rs;
    int base[NI], h0[NI], w0[NI];
    uint32_t off[NI];
    bool ok[NI];
    int kc[NI];                 // element offset inside the K tile of the (swizzled) source chunk this lane fetches for row group j
    int r, s, c0;
  };
  DEV void locate(State& st) const {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      int hi, wi;
      bool v = st.base[j] >= 0 && st.r < g.R;
      if (DGRAD) {
        int hh = st.h0[j] - st.r, ww = st.w0[j] - st.s;
        if (g.stride == 1) { hi = hh; wi = ww; }
        else if (g.stride == 2) { v = v && ((hh | ww) & 1) == 0; hi = hh >> 1; wi = ww >> 1; }
        else { v = v && hh % g.stride == 0 && ww % g.stride == 0; hi = hh / g.stride; wi = ww / g.stride; }
        v = v && hh >= 0 && ww >= 0;
      } else {
        hi = st.h0[j] + st.r; wi = st.w0[j] + st.s;
      }
      v = v && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
      st.ok[j] = v;
      st.off[j] = (uint32_t)(st.base[j] + hi * g.sH + wi * g.sW + st.c0 + st.kc[j]) * 2u;
    }
  }
  DEV void init(State& st, int row0, int wave, int lane, int t_begin) const {
    st.rs = make_rsrc(ptr, bytes);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int trow = (wave * NI + j) * 8 + (lane >> 3);        // row inside the tile
      st.kc[j] = ((lane & 7) ^ ((trow >> 1) & 7)) * EPC;
      const int row = row0 + trow;
      if (row < g.rows) {
        uint32_t n = fd_div(row, g.div_hw);
        uint32_t rem = row - n * g.div_hw.d;
        uint32_t rh = fd_div(rem, g.div_w);
        uint32_t rw = rem - rh * g.div_w.d;
        st.base[j] = n * g.sN;
        if (DGRAD) { st.h0[j] = rh + g.pad; st.w0[j] = rw + g.padw; }
        else { st.h0[j] = rh * g.stride - g.pad; st.w0[j] = rw * g.stride - g.padw; }
      } else {
        st.base[j] = -1; st.h0[j] = 0; st.w0[j] = 0;
      }
    }
    int k0 = t_begin * WIDE_BK;
    int rs = k0 / g.C;
    st.c0 = k0 - rs * g.C;
    st.r = rs / g.S;
    st.s = rs - st.r * g.S;
    locate(st);
  }
  DEV void issue(State& st, char* lds, int wave) const {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const bool v = st.ok[j] && st.c0 + st.kc[j] < g.C;
      buf_load16_lds(st.rs, v ? st.off[j] : OOB_OFF, lds + (wave * NI + j) * 1024);
      st.off[j] += 128;
    }
    st.c0 += WIDE_BK;
    if (st.c0 >= g.C) {          // next window position (block-uniform branch)
      st.c0 = 0;
      if (++st.s == g.S) { st.s = 0; ++st.r; }
      locate(st);
    }
  }
  DEV static int frag_off(int x0, int ks, int lane) {
    const int r = x0 + (lane & 31), c = ks * 2 + (lane >> 5);
    return r * 128 + ((c ^ ((r >> 1) & 7)) << 4);
  }
  DEV static bf16x8 frag_at(const char* p) {
    Chunk16 ch;
    ch.u = *(const u32x4*)p;
    return ch.h;
  }
};

template <class L> struct WideIssue {
  DEV static void go(const L& l, typename L::State& st, char* lds, int wave, int lane, int x0) { l.issue(st, lds, wave, lane, x0); }
};
template <int ROWS, bool D> struct WideIssue<WideKC<ROWS, D>> {
  typedef WideKC<ROWS, D> L;
  DEV static void go(const L& l, typename L::State& st, char* lds, int wave, int, int) { l.issue(st, lds, wave); }
};

// BM x BN tile, WAVES_M x WAVES_N waves of WM x WN each, x KG k-groups = 8 waves
template <int BM_, int BN_, int WM_, int WN_, int KG_>
struct WideCfg {
  static constexpr int BM = BM_, BN = BN_, BK = WIDE_BK, WM = WM_, WN = WN_, KG = KG_;
  static constexpr int WAVES_M = BM / WM, WAVES_N = BN / WN;
  static constexpr int RM = WM / 32, RN = WN / 32;
  static_assert(WAVES_M * WAVES_N * KG == WIDE_NW, "8 waves per workgroup");
  static_assert(WM % 64 == 0, "the epilogue walks the tile in 64-row passes");
  static constexpr int KS = (WIDE_BK / 16) / KG;               // k-steps of a K tile handled by one wave
  // epilogue: passes of 64 tile rows through an f32 staging image
  static constexpr int PR = 64;
  static constexpr int EPI_PITCH = BN * 4 + 16;
  static constexpr int EPI_BYTES = PR * EPI_PITCH;
  static constexpr int CPRE = BN / 8;                          // 8-column chunks per row
  static constexpr int RPSE = 512 / CPRE;                      // rows per sweep of the 512 threads
  static constexpr int ITER = PR / RPSE;                       // rows of a pass handled by one thread
  static constexpr int XCHG_BYTES = KG == 2 ? BM * BN * 4 : 0;
  static constexpr int RED_BYTES = RPSE * CPRE * 16 * 4;
};

// Fused epilogue for the 8-wave kernels. EPI selects which of clite_epilogue's features are compiled in (the flags a launch does not use
// must not cost scalar branches per row: igemm.h's note on igemm_epilogue_plain):
//   0 generic: bias, activation, pre-activation store, activation derivative, dropout, residual, f32 / bf16 store, column statistics
//   1 BatchNorm backward (conv dgrad): relu' mask (before or after the residual), residual, bf16 / f32 store, (sum v, sum v*(bn_y - mean))
//   2 plain: bias, bf16 store, column statistics
// and, round 4, the four forms BERT's linears launch 84 times per step, fixed at compile time like 2 (the run-time form's ~15 wave-uniform flag
// tests per row are scheduling barriers: one row's loads, erf arithmetic, conversions and stores issue back to back instead of interleaving
// with the other rows' — with ONE 8-wave workgroup per CU nothing else hides that; profiles/r2_epilogue_forms.txt: + 15 us on the GELU launches):
//   3 FFN1 forward: bias, pre-activation store, GELU, bf16 store           4 FFN2 input gradient: acc * GELU'(aux), bf16 store, column sums (bias gradient)
//   5 output projections forward: bias, dropout, residual, bf16 store      6 input gradients into a residual stream: + residual, bf16 store
template <int EPI> struct WideEpiForm {
  static constexpr bool RT = EPI == 0, BN = EPI == 1;              // RT: every feature decided by run-time flags
  static constexpr bool PREACT = EPI == 3, GELU = EPI == 3, DGELU = EPI == 4, DROP = EPI == 5, RES = EPI == 5 || EPI == 6;
  static constexpr bool BIAS = EPI == 0 || EPI == 1 || EPI == 2 || EPI == 3 || EPI == 5;          // (1: the folded BatchNorm backward's constant row)
  static constexpr bool MAY_STATS = EPI == 0 || EPI == 1 || EPI == 2 || EPI == 4;
};

template <class CFG, int EPI>
DEV void wide_epilogue(f32x16 (&acc)[CFG::RM][CFG::RN], const Epilogue& ep, const RowMap& rm, char* smem, int M, int N, int m0, int n0,
                       int tid, int lane, int kg, int wm0, int wn0) {
  constexpr int RM = CFG::RM, RN = CFG::RN;
  typedef bf16 T;
  typedef WideEpiForm<EPI> F;
  if constexpr (CFG::KG == 2) {
    // group 1 -> group 0: element (i, j, r) of thread t at ((i*RN + j)*16 + r)*256 + t
    float* xchg = (float*)smem;
    const int t = tid & 255;
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) xchg[((i * RN + j) * 16 + r) * 256 + t] = acc[i][j][r];
    }
    lds_barrier();
    if (kg == 0) {
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += xchg[((i * RN + j) * 16 + r) * 256 + t];
    }
    lds_barrier();
  }
  if (EPI == 0 && ep.atomic) {
    if (kg == 0) {
      float* out = (float*)ep.out;
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) {
          const int col = n0 + wn0 + j * 32 + (lane & 31);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < M && col < N) atomic_add_f32(out + (size_t)row * ep.ldc + col, ep.alpha * acc[i][j][r]);
          }
        }
    }
    return;
  }

  // what this launch's rows do: compile-time constants in the specialised forms, wave-uniform run-time flags in forms 0 / 1
  const bool has_preact = F::PREACT || (F::RT && ep.preact != nullptr);
  const int act = F::GELU ? (int)ACT_GELU : (F::RT ? ep.act : (int)ACT_NONE);
  const bool has_bits = F::BN && ep.relu_bits != nullptr;
  const bool has_aux = F::DGELU || ((F::RT || F::BN) && !has_bits && ep.dact_aux != nullptr);
  const int dact = F::DGELU ? 2 : (F::BN ? 1 : ep.dact);
  const bool has_drop = F::DROP || (F::RT && ep.drop_p > 0.f);
  const bool has_res = F::RES || ((F::RT || F::BN) && ep.residual != nullptr);
  const bool has_y = F::BN && ep.bn_y != nullptr;
  const bool mask_after = F::BN && ep.mask_after_residual != 0;
  const bool to_f32 = (F::RT || F::BN) && ep.out_f32 != 0;
  const bool stats = F::MAY_STATS && ep.colsum != nullptr;
  const bool sums_only = F::DGELU || ep.colsum_rows == 1;          // column sums without the sums of squares (a bias gradient)

  constexpr int CPRE = CFG::CPRE, RPSE = CFG::RPSE, ITER = CFG::ITER, PITCH = CFG::EPI_PITCH;
  const int ecol = (tid % CPRE) * 8, erow0 = tid / CPRE;
  const int gcol = n0 + ecol;
  const bool colok = gcol < N;
  float csum[8], csq[8], bias[8], bn_mean[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { csum[e] = 0.f; csq[e] = 0.f; bias[e] = 0.f; bn_mean[e] = 0.f; }
  if (F::BIAS && colok && ep.bias) {
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = ep.bias[gcol + e];
  }
  if (has_y && colok) {
    for (int r = 0; r < ep.bn_replicas; ++r)
#pragma unroll
      for (int e = 0; e < 8; ++e) bn_mean[e] += ep.bn_stats[(size_t)r * ep.bn_rstride + gcol + e];
#pragma unroll
    for (int e = 0; e < 8; ++e) bn_mean[e] *= ep.bn_inv_count;
  }
  float keep_scale = 1.f;
  uint64_t drop_seed = 0;
  uint32_t drop_site = 0;
  if (has_drop) {
    keep_scale = 1.0f / (1.0f - ep.drop_p);
    drop_seed = ep.drop_seed; drop_site = ep.drop_site;
    seed_resolve(drop_seed, drop_site);
  }

  for (int pass = 0; pass < CFG::BM / CFG::PR; ++pass) {
    // the operands of all rows this thread writes in this pass are requested before the accumulators go through LDS
    Raw8<T> pa[ITER], py[ITER], pr[ITER];
    uint32_t pb[ITER];           // packed relu' bits (clite_epilogue.relu_bits; BatchNorm-backward form only)
    uint32_t gix[ITER];
    bool okr[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int grow = m0 + pass * CFG::PR + erow0 + it * RPSE;
      okr[it] = colok && grow < M;
      gix[it] = okr[it] ? (uint32_t)(map_row(rm, grow) * ep.ldc + gcol) : 0u;
      pb[it] = 0xFFu;
      if (EPI != 2 && EPI != 3 && okr[it]) {
        if (has_bits) pb[it] = ep.relu_bits[gix[it] >> 3];
        if (has_aux) pa[it].ld((const T*)ep.dact_aux + gix[it]);
        if (has_y) py[it].ld((const T*)ep.bn_y + gix[it]);
        if (has_res) pr[it].ld((const T*)ep.residual + gix[it]);
      }
    }
    // the waves that own this pass's 64 rows stage them (static accumulator indices: half h of a wave's rows = acc[2h], acc[2h+1])
#pragma unroll
    for (int h = 0; h < RM / 2; ++h) {
      if (kg == 0 && wm0 + h * 64 == pass * CFG::PR) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
              const int col = wn0 + j * 32 + (lane & 31);
              *(float*)(smem + row * PITCH + col * 4) = acc[h * 2 + ii][j][r];
            }
      }
    }
    lds_barrier();
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      if (!okr[it]) continue;
      const int rr = erow0 + it * RPSE;
      const float* src = (const float*)(smem + rr * PITCH + ecol * 4);
      const f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 4);
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      const size_t gidx = gix[it];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * ep.alpha + bias[e];
      if (has_preact) store8((T*)ep.preact + gidx, v);
      if (act == ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = relu_f(v[e]);
      } else if (act == ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_t<T>(v[e]);
      } else if (act == ACT_TANH) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
      }
      float dfac[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) dfac[e] = 1.f;
      if (has_bits) {
#pragma unroll
        for (int e = 0; e < 8; ++e) dfac[e] = (pb[it] >> e) & 1u ? 1.f : 0.f;
      } else if (has_aux) {
        float av[8];
        pa[it].get(av);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = av[e];
          dfac[e] = dact == 1 ? (a > 0.f ? 1.f : 0.f) : dact == 2 ? gelu_grad_t<T>(a) : (1.f - a * a);
        }
      }
      if ((has_bits || has_aux) && !mask_after) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= dfac[e];
      }
      if (has_drop) {
        float u[8];
        dropout_uniform8(drop_seed, drop_site, gidx, u);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = u[e] >= ep.drop_p ? v[e] * keep_scale : 0.f;
      }
      if (has_res) {
        float rv[8];
        pr[it].get(rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
      }
      if ((has_bits || has_aux) && mask_after) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= dfac[e];
      }
      if (to_f32) {
        store8((float*)ep.out + gidx, v);
      } else {
        store8((T*)ep.out + gidx, v);
        if (F::MAY_STATS) round8_bf16(v);   // statistics of what was stored
      }
      if (has_y) {
        float yv[8];
        py[it].get(yv);
#pragma unroll
        for (int e = 0; e < 8; ++e) { csum[e] += v[e]; csq[e] += v[e] * (yv[e] - bn_mean[e]); }
      } else if (stats) {
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[e] += v[e];
        if (!F::DGELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) csq[e] += v[e] * v[e];
        }
      }
    }
    lds_barrier();
  }

  if (stats) {
    // threads sharing a column chunk are tid, tid + CPRE, ...: fold them through LDS, one atomic per column into replica blockIdx.x % R
    float* crep = ep.colsum + (ep.colsum_replicas > 1 ? (size_t)(blockIdx.x % ep.colsum_replicas) * ep.colsum_stride : 0);
    float* red = (float*)smem;                      // [RPSE][CPRE*16]
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[erow0 * (CPRE * 16) + (tid % CPRE) * 16 + e] = csum[e];
      red[erow0 * (CPRE * 16) + (tid % CPRE) * 16 + 8 + e] = csq[e];
    }
    lds_barrier();
    for (int idx = tid; idx < CPRE * 16; idx += 512) {
      float s = 0.f;
      for (int r = 0; r < RPSE; ++r) s += red[r * (CPRE * 16) + idx];
      const int chunk = idx / 16, e = idx % 16;
      const int col = n0 + chunk * 8 + (e & 7);
      if (col < N && (e < 8 || !sums_only)) atomic_add_f32(crep + (e >= 8 ? N : 0) + col, s);
    }
  }
}

// BatchNorm-backward epilogue of the 8-wave kernels with its operands requested EARLY. One 8-wave workgroup per CU has nothing else on the
// CU to cover an epilogue that waits on HBM: with the per-pass requests of wide_epilogue<CFG, 1> every 64-row pass paid a cold round trip
// (3 x 3 256 -> 256 dgrad at 14 x 14: 73.7 us against 45 for the plain store, profiles/r2_layers.txt). Here the first pass's operands
// (BatchNorm input, residual, packed relu' bits) are requested BEFORE the main loop — their latency hides behind the first K tiles — and all
// later passes' at the start of the epilogue. Tiles whose rows fit the register budget only (EARLY: passes x rows per thread <= 8).
template <class CFG>
struct WideBnOperands {
  typedef bf16 T;
  static constexpr int NPASS = CFG::BM / CFG::PR, ITER = CFG::ITER, ROWS = NPASS * ITER;
  static constexpr bool EARLY = ROWS <= 8;
  Raw8<T> py[ROWS], pr[ROWS], pa[ROWS];
  uint32_t pb[ROWS], gix[ROWS];
  bool okr[ROWS];
  DEV void request(int pass, const Epilogue& ep, const RowMap& rm, int M, int N, int m0, int n0, int tid) {
    const int ecol = (tid % CFG::CPRE) * 8, erow0 = tid / CFG::CPRE;
    const int gcol = n0 + ecol;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int q = pass * ITER + it;
      const int grow = m0 + pass * CFG::PR + erow0 + it * CFG::RPSE;
      okr[q] = gcol < N && grow < M;
      gix[q] = okr[q] ? (uint32_t)(map_row(rm, grow) * ep.ldc + gcol) : 0u;
      pb[q] = 0xFFu;
      if (okr[q]) {
        if (ep.relu_bits) pb[q] = ep.relu_bits[gix[q] >> 3];
        else if (ep.dact_aux) pa[q].ld((const T*)ep.dact_aux + gix[q]);
        if (ep.bn_y) py[q].ld((const T*)ep.bn_y + gix[q]);
        if (ep.residual) pr[q].ld((const T*)ep.residual + gix[q]);
      }
    }
  }
};

template <class CFG>
DEV void wide_epilogue_bn(f32x16 (&acc)[CFG::RM][CFG::RN], WideBnOperands<CFG>& ops, const Epilogue& ep, const RowMap& rm, char* smem, int M, int N,
                          int m0, int n0, int tid, int lane, int kg, int wm0, int wn0) {
  constexpr int RM = CFG::RM, RN = CFG::RN, NPASS = CFG::BM / CFG::PR;
  typedef bf16 T;
  if constexpr (CFG::KG == 2) {
    float* xchg = (float*)smem;
    const int t = tid & 255;
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) xchg[((i * RN + j) * 16 + r) * 256 + t] = acc[i][j][r];
    }
    lds_barrier();
    if (kg == 0) {
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += xchg[((i * RN + j) * 16 + r) * 256 + t];
    }
    lds_barrier();
  }
  constexpr int CPRE = CFG::CPRE, RPSE = CFG::RPSE, ITER = CFG::ITER, PITCH = CFG::EPI_PITCH;
  const int ecol = (tid % CPRE) * 8, erow0 = tid / CPRE;
  const int gcol = n0 + ecol;
  const bool colok = gcol < N;
  float csum[8], csq[8], bn_mean[8], bias[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { csum[e] = 0.f; csq[e] = 0.f; bn_mean[e] = 0.f; bias[e] = (colok && ep.bias) ? ep.bias[gcol + e] : 0.f; }
  if (colok && ep.bn_y) {
    for (int r = 0; r < ep.bn_replicas; ++r)
#pragma unroll
      for (int e = 0; e < 8; ++e) bn_mean[e] += ep.bn_stats[(size_t)r * ep.bn_rstride + gcol + e];
#pragma unroll
    for (int e = 0; e < 8; ++e) bn_mean[e] *= ep.bn_inv_count;
  }
#pragma unroll
  for (int pass = 1; pass < NPASS; ++pass) ops.request(pass, ep, rm, M, N, m0, n0, tid);       // (pass 0 went out before the main loop)
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
    for (int h = 0; h < RM / 2; ++h) {
      if (kg == 0 && wm0 + h * 64 == pass * CFG::PR) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
              const int col = wn0 + j * 32 + (lane & 31);
              *(float*)(smem + row * PITCH + col * 4) = acc[h * 2 + ii][j][r];
            }
      }
    }
    lds_barrier();
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int q = pass * ITER + it;
      if (!ops.okr[q]) continue;
      const int rr = erow0 + it * RPSE;
      const float* src = (const float*)(smem + rr * PITCH + ecol * 4);
      const f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 4);
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      float msk[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[e] = v[e] * ep.alpha + bias[e]; msk[e] = 1.f; }
      if (ep.relu_bits) {
#pragma unroll
        for (int e = 0; e < 8; ++e) msk[e] = (ops.pb[q] >> e) & 1u ? 1.f : 0.f;
      } else if (ep.dact_aux) {
        float av[8];
        ops.pa[q].get(av);
#pragma unroll
        for (int e = 0; e < 8; ++e) msk[e] = av[e] > 0.f ? 1.f : 0.f;
      }
      if (!ep.mask_after_residual) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= msk[e];
      }
      if (ep.residual) {
        float rv[8];
        ops.pr[q].get(rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
      }
      if (ep.mask_after_residual) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= msk[e];
      }
      if (ep.out_f32) {
        store8((float*)ep.out + ops.gix[q], v);
      } else {
        store8((T*)ep.out + ops.gix[q], v);
        round8_bf16(v);   // statistics of what was stored
      }
      if (ep.bn_y) {
        float yv[8];
        ops.py[q].get(yv);
#pragma unroll
        for (int e = 0; e < 8; ++e) { csum[e] += v[e]; csq[e] += v[e] * (yv[e] - bn_mean[e]); }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) { csum[e] += v[e]; csq[e] += v[e] * v[e]; }
      }
    }
    lds_barrier();
  }
  if (ep.colsum) {
    float* crep = ep.colsum + (ep.colsum_replicas > 1 ? (size_t)(blockIdx.x % ep.colsum_replicas) * ep.colsum_stride : 0);
    float* red = (float*)smem;                      // [RPSE][CPRE*16]
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[erow0 * (CPRE * 16) + (tid % CPRE) * 16 + e] = csum[e];
      red[erow0 * (CPRE * 16) + (tid % CPRE) * 16 + 8 + e] = csq[e];
    }
    lds_barrier();
    for (int idx = tid; idx < CPRE * 16; idx += 512) {
      float sacc = 0.f;
      for (int r = 0; r < RPSE; ++r) sacc += red[r * (CPRE * 16) + idx];
      const int chunk = idx / 16, e = idx % 16;
      const int col = n0 + chunk * 8 + (e & 7);
      if (col < N && (e < 8 || ep.colsum_rows != 1)) atomic_add_f32(crep + (e >= 8 ? N : 0) + col, sacc);
    }
  }
}

template <class CFG, class LA, class LB, int NSTAGE, int EPI>
__global__ __launch_bounds__(512) void igemm_wide_kernel(LA la, LB lb, Epilogue ep, RowMap rm, int M, int N, int ktiles, int ktiles_per_split, int xsplits) {
  constexpr int BM = CFG::BM, BN = CFG::BN;
  constexpr int RM = CFG::RM, RN = CFG::RN, KS = CFG::KS;
  constexpr int STAGE = LA::BYTES + LB::BYTES;
  constexpr int RING = NSTAGE * STAGE;
  constexpr int E1 = CFG::EPI_BYTES > CFG::XCHG_BYTES ? CFG::EPI_BYTES : CFG::XCHG_BYTES;
  constexpr int E2 = E1 > CFG::RED_BYTES ? E1 : CFG::RED_BYTES;
  constexpr int SMEM = RING > E2 ? RING : E2;
  constexpr int LOADS_PER_TILE = LA::NI + LB::NI;            // per wave
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(1024))) char smem[SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = wave_uniform(tid >> 6);
  const int kg = wave / (CFG::WAVES_M * CFG::WAVES_N);                 // k-group (0 when KG == 1)
  const int w = wave - kg * (CFG::WAVES_M * CFG::WAVES_N);
  const int wm0 = (w / CFG::WAVES_N) * CFG::WM;
  const int wn0 = (w % CFG::WAVES_N) * CFG::WN;

  const int tiles_n = (N + BN - 1) / BN;
  int wg, zsplit;
  if (xsplits > 0) {
    // split-K with >= 8 splits (weight gradients): all output tiles of one K range run on ONE XCD (igemm_dma.h)
    const int ntile = ((M + BM - 1) / BM) * tiles_n;
    const int j = blockIdx.x >> 3;
    zsplit = (blockIdx.x & 7) + 8 * (j / ntile);
    wg = j % ntile;
    if (zsplit >= xsplits) return;
  } else {
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, xq = nwg >> 3, xr = nwg & 7;
    wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
    zsplit = blockIdx.z;
  }
  const int tm = wg / tiles_n;
  const int tn = wg - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int t_begin = zsplit * ktiles_per_split;
  int t_end = t_begin + ktiles_per_split;
  if (t_end > ktiles) t_end = ktiles;

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

  // loop-invariant LDS offsets of this lane's MFMA fragments: k-steps kg*KS .. kg*KS + KS - 1 of every K tile
  int aoff[RM][KS], boff[RN][KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int i = 0; i < RM; ++i) aoff[i][ks] = LA::frag_off(wm0 + i * 32, kg * KS + ks, lane);
#pragma unroll
    for (int j = 0; j < RN; ++j) boff[j][ks] = LB::frag_off(wn0 + j * 32, kg * KS + ks, lane);
  }

  // BatchNorm-backward form: the epilogue's first-pass operands are requested now, ahead of the prologue DMA (vmcnt counts in issue order, so
  // the first K tile's wait also covers them: their HBM latency overlaps the ring fill instead of opening the epilogue)
  constexpr bool EARLY_BN = EPI == 1 && WideBnOperands<CFG>::EARLY;
  WideBnOperands<CFG> bnops;
  if constexpr (EARLY_BN) bnops.request(0, ep, rm, M, N, m0, n0, tid);
#pragma unroll
  for (int pz = 0; pz < NSTAGE - 1; ++pz) {
    if (t_begin + pz < t_end) {
      WideIssue<LA>::go(la, sa, smem + pz * STAGE, wave, lane, m0);
      WideIssue<LB>::go(lb, sb, smem + pz * STAGE + LA::BYTES, wave, lane, n0);
    }
  }
  // (Measured and rejected, round 3: `s_setprio 1` for waves 4-7 — the static priority for the later-dispatched half that MI355X_MICROARCH.md
  // suggests for two same-program waves per SIMD — step 16.88 vs 16.90 ms, 3x3 128 -> 128 conv 62.3 -> 63.2 us.)
  int buf = 0;
  for (int t = t_begin; t < t_end; ++t) {
    // tile t has landed once at most min(NSTAGE-2, tiles after t) younger tiles are still outstanding
    const int after = t_end - 1 - t;
    if (NSTAGE >= 4 && after >= 2) wait_vmcnt<2 * LOADS_PER_TILE>();
    else if (NSTAGE >= 3 && after >= 1) wait_vmcnt<LOADS_PER_TILE>();
    else wait_vmcnt<0>();
    barrier_raw();            // everyone's part of tile t has landed; everyone is done reading the buffer tile t+NSTAGE-1 overwrites
    const char* abuf = smem + buf * STAGE;
    const char* bbuf = abuf + LA::BYTES;
    // first k-step's fragment reads go out before the next tile's DMA is issued (the issue overlaps the LDS latency)
    // (diagnostic variants, `make variant VAR_EXTRA=-DCLITE_ABLATE=n`, tools/probe_rows768.py, profiles/r5_wide_ablation.txt: 1 = no fragment reads / MFMAs (the
    // memory side alone), 2 = no DMA (reads + MFMAs + barriers), 3 = neither DMA nor fragment reads (MFMAs + barriers). Round 5, 3840 x 768 x 3072 on the
    // 128 x 128 tile: whole 32.5 us, 1: 22.8, 2: 24.9, 3: 23.2 — two sides of equal length that overlap to 70 %; the MFMA side is the matrix pipe itself
    // (53 cycles per 32x32x16 instruction with two waves per SIMD), the memory side one CU's LDS-DMA ingest (86 GB/s, the same with 90 tiles as with 180).
    // Tried on top of that and removed: a 4-stage ring (32.2 us); fragments of k-step s + 1 read under the MFMAs of k-step s across the tile boundary, with
    // the compiler's waits (33.0) and with hand-counted s_waitcnt lgkmcnt on untracked reads (32.1; compute side alone 23.9) — and +0.15 ms in the step.)
    bf16x8 af0[RM], bf0[RN];
#pragma unroll
    for (int i = 0; i < RM; ++i) af0[i] = LA::frag_at(abuf + aoff[i][0]);
#pragma unroll
    for (int j = 0; j < RN; ++j) bf0[j] = LB::frag_at(bbuf + boff[j][0]);
#if CLITE_ABLATE != 2 && CLITE_ABLATE != 3
    if (t + NSTAGE - 1 < t_end) {
      int nb = buf + NSTAGE - 1; if (nb >= NSTAGE) nb -= NSTAGE;
      WideIssue<LA>::go(la, sa, smem + nb * STAGE, wave, lane, m0);
      WideIssue<LB>::go(lb, sb, smem + nb * STAGE + LA::BYTES, wave, lane, n0);
    }
#endif
#pragma unroll
    for (int ks = 0; ks < (CLITE_ABLATE == 1 ? 0 : KS); ++ks) {
      bf16x8 af[RM], bfr[RN];
#pragma unroll
      for (int i = 0; i < RM; ++i) af[i] = (ks == 0 || CLITE_ABLATE == 3) ? af0[i] : LA::frag_at(abuf + aoff[i][ks]);
#pragma unroll
      for (int j = 0; j < RN; ++j) bfr[j] = (ks == 0 || CLITE_ABLATE == 3) ? bf0[j] : LB::frag_at(bbuf + boff[j][ks]);
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) acc[i][j] = mfma32_bf16(af[i], bfr[j], acc[i][j]);
    }
    if (++buf == NSTAGE) buf = 0;
  }
  lds_barrier();          // no DMA outstanding; every wave is past its last fragment read: the LDS is free for the epilogue
  if constexpr (EARLY_BN) wide_epilogue_bn<CFG>(acc, bnops, ep, rm, smem, M, N, m0, n0, tid, lane, kg, wm0, wn0);
  else wide_epilogue<CFG, EPI>(acc, ep, rm, smem, M, N, m0, n0, tid, lane, kg, wm0, wn0);
}

}  // namespace clite
#endif  // CLITE_IGEMM_WIDE_H
