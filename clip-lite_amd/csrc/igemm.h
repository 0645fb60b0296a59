// Implicit-GEMM tile engine for gfx950: C[M,N] (+)= Aop[M,K] * Bop[N,K]^T, bf16 in, fp32 accumulate
// on v_mfma_f32_32x32x16_bf16.
//
// One engine serves every dense contraction of the CLIP-Lite train step:
//   * conv forward  (reference: torchvision ResNet convs behind encoder.py:36-65): A = NHWC activations
//     gathered im2col-style, B = weights [Cout][R*S*Cin]
//   * conv dgrad: A = dY gathered with transposed-conv geometry, B = weights read k-strided ("XC" image)
//   * conv/linear wgrad: both operands are contracted over their slow (pixel / token) dimension, so both
//     LDS images are [k][x] and the MFMA fragments come from ds_read_b64_tr_b16 transposed reads
//   * every nn.Linear of BERT (encoder.py:165-196) and of the MI heads / priors (loss.py:12-53).
//
// Operand images in LDS:
//   KC (k-contiguous in memory): [ROWS][BK] bf16, row pitch BK*2+16 B  -> conflict-free ds_read_b128 fragments
//   XC (x-contiguous in memory, contraction index is the slow one): [BK][ROWS] bf16, pitch ROWS*2+64 B
//       -> conflict-free ds_read_b64_tr_b16 (a 32-lane half covers 4 k-rows x 64 B on disjoint banks)
// Staging is global -> VGPR (buffer loads: out-of-range = 0, which implements im2col padding, ragged
// tile edges and K tails without branches) -> ds_write_b128, double-buffered, one barrier per K tile;
// the loads of tile t+1 are issued before the MFMAs of tile t and written after them.
#ifndef CLITE_IGEMM_H
#define CLITE_IGEMM_H
#include "intrin.h"
#include "rng.h"
#include "vec.h"
#include "clite.h"

#ifndef EPI_STAMP
#define EPI_STAMP(i) do {} while (0)
#endif
#ifndef BN_EPI_PHASE
#define BN_EPI_PHASE(i) do {} while (0)
#endif
// tuning constants of the BatchNorm-backward epilogue (igemm_epilogue_bn); `make variant VAR_EXTRA=-D...` builds A/B them on one box
#ifndef CLITE_BN_AHEAD
#define CLITE_BN_AHEAD 4           // rows of epilogue operands in flight ahead of the row being written (run-time form)
#endif
#ifndef CLITE_BN_AHEAD_FORM
#define CLITE_BN_AHEAD_FORM 4      // ... in the specialised forms (fewer registers per row: no tensor-form mask)
#endif
#ifndef CLITE_BN_EARLY
#define CLITE_BN_EARLY 4           // rows of epilogue operands igemm_dma_bn_kernel requests BEFORE a tile's main loop (specialised forms; 0: none)
#endif
#ifndef CLITE_BN_HALF
#define CLITE_BN_HALF 0            // 1: half-tile staging + three workgroups per CU (measured: no gain)
#endif

namespace clite {

struct FastDiv {  // q = n / d for n < 2^31, d >= 1 (host fills; see fastdiv_make)
  uint32_t mul, shift, d;
};
DEV uint32_t fd_div(uint32_t n, FastDiv f) { return (umulhi32(n, f.mul) + n) >> f.shift; }

// Geometry of a gathered NHWC tensor seen through a conv window. Element strides are explicit so the
// 7x7 stem can be expressed on a pre-padded [N][Hp][Wp][4] image as a 7x1 window over 32 "virtual"
// channels (8 adjacent pixels x 4 channels are contiguous in memory).
struct ConvGeom {
  int H, W, C;        // spatial extent and channel count (k-inner) of the gathered tensor
  int sN, sH, sW;     // element strides of the gathered tensor
  int RH, RW;         // spatial extent of the row space (output pixels for fwd/wgrad, input pixels for dgrad)
  int R, S;           // window
  int stride, pad, padw;   // pad: rows (h), padw: columns (w); equal except in the parity-class dgrads of strided convs
  int rows;           // N * RH * RW
  int concat;         // 1: the "window" is a K-concatenation of R tensors laid out back to back (clite_conv_dgrad_bnfold: H = R slots, W = rows) —
                      // a short-K 1 x 1 problem for the launch policies, not a sliding window
  FastDiv div_hw, div_w;  // by RH*RW and by RW
};

// ------------------------------------------------------------------------------------------------
// KC gather loader: rows = pixels of the row space, k = (r, s, c) with c contiguous.
// DGRAD=false: hi = rh*stride - pad + r.   DGRAD=true: hi = (rh + pad - r) / stride when divisible.
template <typename T, int ROWS, int BK, bool DGRAD>
struct GatherKC {
  static constexpr int EPC = 16 / (int)sizeof(T);    // elements per 16-B chunk
  static constexpr int CPR = BK / EPC;               // 16-B chunks per row
  static constexpr int NCH = ROWS * CPR / 256;       // chunks per thread
  static constexpr int PITCH = BK * (int)sizeof(T) + 16;
  static constexpr int BYTES = ROWS * PITCH;
  static constexpr bool XC = false;
  const void* ptr;
  uint32_t bytes;
  ConvGeom g;

  struct State {
    rsrc_t rs;
    int base[NCH], h0[NCH], w0[NCH];  // base = n*sN, or -1 when the row is out of range
    int kc8;                          // element offset of this thread's chunk inside the K tile
    int r, s, c0;
  };
  DEV void init(State& st, int row0, int tid, int t_begin) const {
    st.rs = make_rsrc(ptr, bytes);
    st.kc8 = (tid % CPR) * EPC;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int row = row0 + (tid + 256 * i) / CPR;
      if (row < g.rows) {
        uint32_t n = fd_div(row, g.div_hw);
        uint32_t rem = row - n * g.div_hw.d;
        uint32_t rh = fd_div(rem, g.div_w);
        uint32_t rw = rem - rh * g.div_w.d;
        st.base[i] = n * g.sN;
        if (DGRAD) { st.h0[i] = rh + g.pad; st.w0[i] = rw + g.padw; }
        else { st.h0[i] = rh * g.stride - g.pad; st.w0[i] = rw * g.stride - g.padw; }
      } else {
        st.base[i] = -1; st.h0[i] = 0; st.w0[i] = 0;
      }
    }
    int k0 = t_begin * BK;
    int rs = k0 / g.C;
    st.c0 = k0 - rs * g.C;
    st.r = rs / g.S;
    st.s = rs - st.r * g.S;
  }
  DEV void load(State& st, u32x4 (&regs)[NCH]) const {
    int c = st.c0 + st.kc8;
    bool kvalid = c < g.C && st.r < g.R;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int hi, wi;
      bool v = kvalid && st.base[i] >= 0;
      if (DGRAD) {
        int hh = st.h0[i] - st.r, ww = st.w0[i] - st.s;
        if (g.stride == 1) { hi = hh; wi = ww; }
        else if (g.stride == 2) { v = v && ((hh | ww) & 1) == 0; hi = hh >> 1; wi = ww >> 1; }
        else { v = v && hh % g.stride == 0 && ww % g.stride == 0; hi = hh / g.stride; wi = ww / g.stride; }
        v = v && hh >= 0 && ww >= 0;
      } else {
        hi = st.h0[i] + st.r; wi = st.w0[i] + st.s;
      }
      v = v && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
      uint32_t off = v ? (uint32_t)(st.base[i] + hi * g.sH + wi * g.sW + c) * (uint32_t)sizeof(T) : OOB_OFF;
      regs[i] = buf_load16(st.rs, off);
    }
    st.c0 += BK;
    if (st.c0 >= g.C) { st.c0 = 0; if (++st.s == g.S) { st.s = 0; ++st.r; } }
  }
  DEV static void store(char* lds, int tid, const u32x4 (&regs)[NCH]) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int c = tid + 256 * i;
      *(u32x4*)(lds + (c / CPR) * PITCH + (c % CPR) * 16) = regs[i];
    }
  }
  // bf16: MFMA 32x32x16 fragment for the 32-row block starting at x0, k-step ks (16 deep)
  DEV static bf16x8 frag(const char* lds, int x0, int ks, int lane) {
    Chunk16 ch;
    ch.u = *(const u32x4*)(lds + (x0 + (lane & 31)) * PITCH + (ks * 16 + 8 * (lane >> 5)) * 2);
    return ch.h;
  }
  // f32: MFMA 32x32x2 operand (one value per lane), k-step kk (2 deep)
  DEV static float frag32(const char* lds, int x0, int kk, int lane) {
    return *(const float*)(lds + (x0 + (lane & 31)) * PITCH + (kk * 2 + (lane >> 5)) * 4);
  }
};

// ------------------------------------------------------------------------------------------------
// XC strided loader: element (k, x) at ptr[(k0+krow)*ld + rs*Cx + x]; k = (rs, kk) with kk < Ck.
// Serves: weights for dgrad (W[co][r][s][ci]: ld = R*S*Cin, Cx = Cin, Ck = Cout, RS = R*S),
//         linear dgrad (W[n][k]: ld = K, Cx = K, Ck = N, RS = 1), wgrad's dY ([P][Cout]: ld = Cx = Cout, Ck = P).
template <typename T, int COLS, int BK>
struct StridedXC {
  static constexpr int EPC = 16 / (int)sizeof(T);
  static constexpr int CPR = COLS / EPC;
  static constexpr int NCH = BK * CPR / 256;
  static constexpr int RPS = 256 / CPR;              // k-rows per sweep of the block
  static constexpr int PITCH = COLS * (int)sizeof(T) + 64;
  static constexpr int BYTES = BK * PITCH;
  static constexpr bool XC = true;
  const void* ptr;
  uint32_t bytes;
  int ld, Cx, Ck, RS;

  struct State {
    rsrc_t rs_;
    int x, krow0;
    bool xvalid;
    int rs, k0;
  };
  DEV void init(State& st, int x0, int tid, int t_begin) const {
    st.rs_ = make_rsrc(ptr, bytes);
    st.x = x0 + (tid % CPR) * EPC;
    st.xvalid = st.x < Cx;
    st.krow0 = tid / CPR;
    int kk = t_begin * BK;
    st.rs = (RS == 1) ? 0 : kk / Ck;
    st.k0 = kk - st.rs * Ck;
  }
  DEV void load(State& st, u32x4 (&regs)[NCH]) const {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int k = st.k0 + st.krow0 + i * RPS;
      bool v = st.xvalid && k < Ck && st.rs < RS;
      uint32_t off = v ? (uint32_t)(k * ld + st.rs * Cx + st.x) * (uint32_t)sizeof(T) : OOB_OFF;
      regs[i] = buf_load16(st.rs_, off);
    }
    st.k0 += BK;
    if (st.k0 >= Ck && RS > 1) { st.k0 = 0; ++st.rs; }
  }
  DEV static void store(char* lds, int tid, const u32x4 (&regs)[NCH]) {
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      *(u32x4*)(lds + (tid / CPR + i * RPS) * PITCH + (tid % CPR) * 16) = regs[i];
  }
  DEV static bf16x8 frag(const char* lds, int x0, int ks, int lane) {
    // lane 4q+p of each 16-lane group addresses k-row q, columns 4p..4p+3 of a [4 k][16 x] block
    int x = x0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    int k = ks * 16 + 8 * (lane >> 5) + ((lane >> 2) & 3);
    const char* p = lds + k * PITCH + x * 2;
    s16x4 lo = lds_read_tr16(p);
    s16x4 hi = lds_read_tr16(p + 4 * PITCH);
    union { s16x4 v[2]; bf16x8 h; } u;
    u.v[0] = lo; u.v[1] = hi;
    return u.h;
  }
  DEV static float frag32(const char* lds, int x0, int kk, int lane) {
    return *(const float*)(lds + (kk * 2 + (lane >> 5)) * PITCH + (x0 + (lane & 31)) * 4);
  }
};

// ------------------------------------------------------------------------------------------------
// XC gather loader (wgrad's activation operand): k = output pixel p -> (n, ho, wo); x = (r, s, ci).
template <typename T, int COLS, int BK>
struct GatherXC {
  static constexpr int EPC = 16 / (int)sizeof(T);
  static constexpr int CPR = COLS / EPC;
  static constexpr int NCH = BK * CPR / 256;
  static constexpr int RPS = 256 / CPR;
  static constexpr int PITCH = COLS * (int)sizeof(T) + 64;
  static constexpr int BYTES = BK * PITCH;
  static constexpr bool XC = true;
  const void* ptr;
  uint32_t bytes;
  ConvGeom g;   // rows = N*Ho*Wo pixels (the contraction index); RH,RW = Ho,Wo

  struct State {
    rsrc_t rs_;
    int xoff;      // r*sH + s*sW + ci, or -1 if this column is out of range
    int r, s;
    int krow0, p0;
  };
  DEV void init(State& st, int x0, int tid, int t_begin) const {
    st.rs_ = make_rsrc(ptr, bytes);
    int x = x0 + (tid % CPR) * EPC;
    int rs = x / g.C;
    int ci = x - rs * g.C;
    st.r = rs / g.S;
    st.s = rs - st.r * g.S;
    st.xoff = (x < g.R * g.S * g.C) ? ci : -1;
    st.krow0 = tid / CPR;
    st.p0 = t_begin * BK;
  }
  DEV void load(State& st, u32x4 (&regs)[NCH]) const {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int p = st.p0 + st.krow0 + i * RPS;
      bool v = st.xoff >= 0 && p < g.rows;
      uint32_t n = fd_div(p, g.div_hw);
      uint32_t rem = p - n * g.div_hw.d;
      uint32_t ho = fd_div(rem, g.div_w);
      uint32_t wo = rem - ho * g.div_w.d;
      int hi = (int)ho * g.stride - g.pad + st.r;
      int wi = (int)wo * g.stride - g.padw + st.s;
      v = v && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
      uint32_t off = v ? (uint32_t)((int)n * g.sN + hi * g.sH + wi * g.sW + st.xoff) * (uint32_t)sizeof(T) : OOB_OFF;
      regs[i] = buf_load16(st.rs_, off);
    }
    st.p0 += BK;
  }
  DEV static void store(char* lds, int tid, const u32x4 (&regs)[NCH]) {
    StridedXC<T, COLS, BK>::store(lds, tid, regs);
  }
  DEV static bf16x8 frag(const char* lds, int x0, int ks, int lane) {
    return StridedXC<T, COLS, BK>::frag(lds, x0, ks, lane);
  }
  DEV static float frag32(const char* lds, int x0, int kk, int lane) {
    return StridedXC<T, COLS, BK>::frag32(lds, x0, kk, lane);
  }
};

// ------------------------------------------------------------------------------------------------
// Epilogue description (runtime flags; all branches are block-uniform).
enum { ACT_NONE = CLITE_ACT_NONE, ACT_RELU = CLITE_ACT_RELU, ACT_GELU = CLITE_ACT_GELU, ACT_TANH = CLITE_ACT_TANH };
typedef clite_epilogue Epilogue;   // layout and semantics: include/clite.h

DEV float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
DEV float gelu_grad_f(float x) {
  float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
// bf16 storage: the library erff (two branches, ~30 instructions) is most of a GELU epilogue's time (FFN1 forward / FFN2 dgrad: 48 vs 33 us with
// a plain store). erf by Abramowitz & Stegun 7.1.26 — 1 - (a1 t + ... + a5 t^5) e^(-z^2), t = 1 / (1 + p z), |error| <= 1.5e-7, far below the
// 2^-9 of the stored value — costs one reciprocal, one exponential and six FMAs, and the derivative reuses the exponential (e^(-z^2) with
// z = x / sqrt 2 IS the Gaussian of the density). The f32 parity mode keeps erff.
// (Round 4: constants folded — 1 / (1 + p |x| / sqrt 2), e^(-x^2 / 2) as exp2 of ONE product, the 0.5 inside the coefficients — and the sign
// select replaced by arithmetic: with h = 0.5 erfc(|x| / sqrt 2), x Phi(x) = max(x, 0) - |x| h. 13 / 16 instructions per element instead of 18 / 21:
// at one 8-wave workgroup per CU the GELU launches' epilogue is VALU-bound, 128 elements per thread.)
DEV void gelu_fast_parts(float x, float& half_erfc, float& gauss) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(__fmaf_rn(0.3275911f * 0.70710678118654752f, ax, 1.0f));
  gauss = __builtin_amdgcn_exp2f((x * x) * (-0.5f * 1.4426950408889634f));          // e^(-x^2 / 2)
  const float poly = ((((0.5f * 1.061405429f * t - 0.5f * 1.453152027f) * t + 0.5f * 1.421413741f) * t - 0.5f * 0.284496736f) * t + 0.5f * 0.254829592f) * t;
  half_erfc = poly * gauss;            // 0.5 * erfc(|x| / sqrt 2) = Phi(-|x|)
}
template <typename T> DEV float gelu_t(float x) {
  if constexpr (sizeof(T) == 2) {
    float h, g;
    gelu_fast_parts(x, h, g);
    return __fmaf_rn(-fabsf(x), h, fmaxf(x, 0.f));          // x >= 0: x (1 - h); x < 0: x h  (a NaN stays a NaN through the product)
  } else {
    return gelu_f(x);
  }
}
template <typename T> DEV float gelu_grad_t(float x) {
  if constexpr (sizeof(T) == 2) {
    float h, g;
    gelu_fast_parts(x, h, g);
    const float cdf = 0.5f + copysignf(0.5f - h, x);          // Phi(x) = 1 - h for x >= 0, h for x < 0
    return __fmaf_rn(x * 0.39894228040143268f, g, cdf);
  } else {
    return gelu_grad_f(x);
  }
}

// Optional remap of GEMM rows to rows of the output (and of the residual / aux tensors): GEMM row p = (n, ho, wo) over an
// [N][Ho][Wo] grid lands on pixel (n, ho*stride, wo*stride) of an [N][H][W] tensor. Used by the 1x1 / stride-2 dgrad, which is a
// dense GEMM over the output pixels scattered into every stride-th input pixel.
// on = 2 (BatchNorm-backward epilogue only): the OUTPUT rows are dense, and the RESIDUAL is a compact tensor over every second pixel of the
// [N][H][W] grid the rows run over (clite_epilogue.residual_subsample): div_hw / div_w divide by H * W and W here.
struct RowMap {
  int on;
  FastDiv div_hw, div_w;   // by Ho*Wo and by Wo
  int H, W, stride;
  int off_h, off_w;        // pixel (n, ho*stride + off_h, wo*stride + off_w)
};
// rm.on == 2: row of the compact residual tensor [N][H/2][W/2] that is added to output row `row` of the [N][H][W] grid, or -1 (odd h or w)
DEV int subsampled_row(const RowMap& rm, int row) {
  const uint32_t n = fd_div(row, rm.div_hw);
  const uint32_t rem = row - n * rm.div_hw.d;
  const uint32_t h = fd_div(rem, rm.div_w);
  const uint32_t w = rem - h * rm.div_w.d;
  return ((h | w) & 1u) ? -1 : (int)((n * (uint32_t)(rm.H >> 1) + (h >> 1)) * (uint32_t)(rm.W >> 1) + (w >> 1));
}
DEV size_t map_row(const RowMap& rm, int row) {
  if (rm.on != 1) return (size_t)row;
  uint32_t n = fd_div(row, rm.div_hw);
  uint32_t rem = row - n * rm.div_hw.d;
  uint32_t ho = fd_div(rem, rm.div_w);
  uint32_t wo = rem - ho * rm.div_w.d;
  return ((size_t)n * rm.H + ho * rm.stride + rm.off_h) * rm.W + wo * rm.stride + rm.off_w;
}

template <int BM_, int BN_, int BK_, int WM_, int WN_>
struct TileCfg {
  static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_;
  static constexpr int WAVES_M = BM / WM, WAVES_N = BN / WN;
  static constexpr int RM = WM / 32, RN = WN / 32;
  static_assert(WAVES_M * WAVES_N == 4, "256-thread workgroups");
#ifndef CLITE_EPI_PAD
#define CLITE_EPI_PAD 16          // A/B builds: 0 makes the BatchNorm-backward kernels' whole-tile staging 64 KB instead of 67.6 (what fits beside a 96 KB workgroup on a CU)
#endif
  static constexpr int EPI_PITCH = BN * 4 + CLITE_EPI_PAD;
  static constexpr int EPI_BYTES = WM * EPI_PITCH;
};

// Fused epilogue shared by the register-staged and the LDS-DMA kernels. `smem` must be free (all waves past their last
// fragment read) and at least max(CFG::EPI_BYTES, 16 KB) large.
// HEAVY = the BatchNorm-backward form (clite_epilogue.bn_y / mask_after_residual, operands prefetched two rows at a time). It costs
// ~45 more registers than the plain form, which would take every kernel from 3 to 2 workgroups per CU, so it is a separate
// instantiation used only by the launches that ask for it (conv dgrad inside the ResNet backward).
// Q8 (clite_gemm_nt_fp8's instantiation only): clite_epilogue.fp8_out / fp8_scale / fp8_amax - the e4m3 copy of the stored bf16 value at a
// delayed scale and the call's max |out|, as bn_apply's producer-fused quantiser (resnet_ops.hip).
template <typename T, class CFG, bool HEAVY = false, bool Q8 = false>
DEV void igemm_epilogue(f32x16 (&acc)[CFG::RM][CFG::RN], const Epilogue& ep, const RowMap& rm, char* smem, int M, int N, int m0, int n0,
                        int tid, int lane, int wave, int wm0, int wn0) {
  constexpr int BN = CFG::BN;
  constexpr int RM = CFG::RM, RN = CFG::RN;
  if (ep.atomic) {
    float* out = (float*)ep.out;
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
      for (int j = 0; j < RN; ++j) {
        int col = n0 + wn0 + j * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row < M && col < N) atomic_add_f32(out + (size_t)row * ep.ldc + col, ep.alpha * acc[i][j][r]);
        }
      }
    return;
  }

  constexpr int CPRE = BN / 8;          // 8-column chunks per tile row
  constexpr int RPSE = 256 / CPRE;      // rows per sweep
  const int ecol = (tid % CPRE) * 8;
  const int erow0 = tid / CPRE;
  const float keep_scale = ep.drop_p > 0.f ? 1.0f / (1.0f - ep.drop_p) : 1.0f;
  uint64_t drop_seed = ep.drop_seed;
  uint32_t drop_site = ep.drop_site;
  if (ep.drop_p > 0.f) seed_resolve(drop_seed, drop_site);
  float csum[8], csq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { csum[e] = 0.f; csq[e] = 0.f; }
  float qmax = 0.f;
  bool qnan = false;
  const float qscale = (Q8 && ep.fp8_out) ? ep.fp8_scale[0] : 1.f;

  // per-column constants of this thread's 8 columns
  const int gcol = n0 + ecol;
  const bool colok = gcol < N;
  constexpr int ITER = CFG::WM / RPSE;      // rows of a pass handled by one thread
  static_assert(CFG::WM % RPSE == 0, "tile shape");

  for (int pass = 0; pass < CFG::WAVES_M; ++pass) {
    // the epilogue operands (activation-derivative aux, BatchNorm input, residual) of all rows this thread will write in this pass
    // are requested up front, before the accumulators go through LDS: their latency overlaps the staging instead of being paid
    // once per row
    constexpr int PF = !HEAVY ? 1 : (ITER < 2 ? ITER : 2);       // rows prefetched at a time (register budget: 3 operands x PF x 16 B)
    Raw8<T> pa[PF], py[PF], pr[PF];
    uint32_t gix[PF];       // element offsets (every tensor of the step has < 2^30 elements: gemm.hip fits32)
    bool okr[PF];
    auto prefetch = [&](int it0) {
#pragma unroll
      for (int q = 0; q < PF; ++q) {
        int grow = m0 + pass * CFG::WM + erow0 + (it0 + q) * RPSE;
        okr[q] = colok && grow < M;
        gix[q] = okr[q] ? (uint32_t)(map_row(rm, grow) * ep.ldc + gcol) : 0u;
        if constexpr (HEAVY) {
          if (okr[q]) {
            if (ep.dact_aux) pa[q].ld((const T*)ep.dact_aux + gix[q]);
            if (ep.bn_y) py[q].ld((const T*)ep.bn_y + gix[q]);
            if (ep.residual) pr[q].ld((const T*)ep.residual + gix[q]);
          }
        }
      }
    };
    prefetch(0);
    if (wave / CFG::WAVES_N == pass) {
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            int col = wn0 + j * 32 + (lane & 31);
            *(float*)(smem + row * CFG::EPI_PITCH + col * 4) = acc[i][j][r];
          }
    }
    lds_barrier();
    if (pass == 0) EPI_STAMP(6);
    float bias[8], bn_mean[8];          // per-column constants (L2-resident; reloaded per pass to keep them out of the staging phase's registers)
#pragma unroll
    for (int e = 0; e < 8; ++e) { bias[e] = (colok && ep.bias) ? ep.bias[gcol + e] : 0.f; bn_mean[e] = 0.f; }
    if (HEAVY && colok && ep.bn_y) {
      for (int r = 0; r < ep.bn_replicas; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) bn_mean[e] += ep.bn_stats[(size_t)r * ep.bn_rstride + gcol + e];
#pragma unroll
      for (int e = 0; e < 8; ++e) bn_mean[e] *= ep.bn_inv_count;
    }
    // fully unrolled: the rows of a thread are independent, so their LDS reads, conversions and stores interleave (one wave per SIMD
    // issues a single row's dependent chain at a fraction of the VALU rate)
#pragma unroll
    for (int it0 = 0; it0 < ITER; it0 += PF) {
      if (it0 > 0) prefetch(it0);
#pragma unroll
      for (int q = 0; q < PF; ++q) {
      if (!okr[q]) continue;
      const int rr = erow0 + (it0 + q) * RPSE;
      const float* src = (const float*)(smem + rr * CFG::EPI_PITCH + ecol * 4);
      f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 4);
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      const size_t gidx = gix[q];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * ep.alpha + bias[e];
      if (ep.preact) store8((T*)ep.preact + gidx, v);
      if (ep.act == ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = relu_f(v[e]);
      } else if (ep.act == ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_t<T>(v[e]);
      } else if (ep.act == ACT_TANH) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
      }
      float dfac[8];
      if (ep.dact_aux) {
        float av[8];
        if constexpr (HEAVY) pa[q].get(av); else load8((const T*)ep.dact_aux + gidx, av);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float a = av[e];
          dfac[e] = ep.dact == 1 ? (a > 0.f ? 1.f : 0.f) : ep.dact == 2 ? gelu_grad_t<T>(a) : (1.f - a * a);
        }
        if (!HEAVY || !ep.mask_after_residual) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= dfac[e];
        }
      }
      if (ep.drop_p > 0.f) {
        float u[8];
        dropout_uniform8(drop_seed, drop_site, gidx, u);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = u[e] >= ep.drop_p ? v[e] * keep_scale : 0.f;
      }
      if (ep.residual) {
        float rv[8];
        if constexpr (HEAVY) pr[q].get(rv); else load8((const T*)ep.residual + gidx, rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
      }
      if (HEAVY && ep.dact_aux && ep.mask_after_residual) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= dfac[e];
      }
      if (ep.out_f32 || sizeof(T) == 4) {
        store8((float*)ep.out + gidx, v);
      } else {
        store8((bf16*)ep.out + gidx, v);
        round8_bf16(v);   // statistics of what was stored
        if constexpr (Q8) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { qmax = fmaxf(qmax, fabsf(v[e])); qnan = qnan || v[e] != v[e]; }
          if (ep.fp8_out) {
            uint32_t w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              // (a NaN passes the clamp untouched and converts to e4m3's NaN, as in fp8_ops.hip)
              const float p0 = v[2 * e] * qscale, q0 = v[2 * e + 1] * qscale;
              const float pp = p0 != p0 ? p0 : fminf(fmaxf(p0, -448.f), 448.f), qq = q0 != q0 ? q0 : fminf(fmaxf(q0, -448.f), 448.f);
              w[e] = cvt2_fp8(pp, qq);
            }
            *(u32x2*)(ep.fp8_out + gidx) = u32x2{w[0] | (w[1] << 16), w[2] | (w[3] << 16)};
          }
        }
      }
      if (HEAVY && ep.bn_y) {
        float yv[8];
        py[q].get(yv);
#pragma unroll
        for (int e = 0; e < 8; ++e) { csum[e] += v[e]; csq[e] += v[e] * (yv[e] - bn_mean[e]); }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) { csum[e] += v[e]; csq[e] += v[e] * v[e]; }
      }
    }
    }
    if (pass == 0) EPI_STAMP(7);
    lds_barrier();
  }

  if constexpr (Q8) {
    if (ep.fp8_amax) {          // one integer atomic max per workgroup on the bit pattern (non-negative floats order like their bits; the quiet-NaN pattern above all)
      uint32_t mb = qnan ? 0x7FC00000u : f32_bits(qmax);
#pragma unroll
      for (int sh = 32; sh >= 1; sh >>= 1) { const uint32_t o = (uint32_t)wave_shfl_xor_i((int)mb, sh); mb = o > mb ? o : mb; }
      uint32_t* red = (uint32_t*)smem;
      if (lane == 0) red[wave] = mb;
      lds_barrier();
      if (tid == 0) {
        uint32_t b = red[0];
        for (int w = 1; w < 4; ++w) b = red[w] > b ? red[w] : b;
        atomic_max_u32((uint32_t*)ep.fp8_amax + (blockIdx.x % CLITE_FP8_AMAX_REPLICAS) * CLITE_FP8_AMAX_STRIDE, b);
      }
      lds_barrier();
    }
  }
  if (ep.colsum) {
    // threads sharing a column chunk are tid, tid+CPRE, ...: fold them through LDS, one atomic per column. The accumulator
    // is replicated (workgroup b adds into replica b % R) because thousands of tiles adding to the same few addresses
    // serialise at the memory side; consumers fold the replicas when they read the statistics.
    float* crep = ep.colsum + (ep.colsum_replicas > 1 ? (size_t)(blockIdx.x % ep.colsum_replicas) * ep.colsum_stride : 0);
    float* red = (float*)smem;                      // [RPSE][CPRE*16]
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[erow0 * (CPRE * 16) + (tid % CPRE) * 16 + e] = csum[e];
      red[erow0 * (CPRE * 16) + (tid % CPRE) * 16 + 8 + e] = csq[e];
    }
    lds_barrier();
    for (int idx = tid; idx < CPRE * 16; idx += 256) {
      float s = 0.f;
      for (int r = 0; r < RPSE; ++r) s += red[r * (CPRE * 16) + idx];
      int chunk = idx / 16, e = idx % 16;
      int col = n0 + chunk * 8 + (e & 7);
      if (col < N && (e < 8 || ep.colsum_rows != 1)) atomic_add_f32(crep + (e >= 8 ? N : 0) + col, s);
    }
  }
}

// Plain form of the epilogue as its own instantiation (EPI = 2): bf16 store of alpha*acc + bias, optionally with the column statistics —
// every conv forward, the QKV projection, plain dgrads. The generic body tests ~15 wave-uniform flags per row, and at one wave per SIMD
// those scalar branches — not the stores, not the LDS reads (both ablated with tools/probe_stamps.py) — are what a row costs: this
// body has none of them (epilogue of the 1x1 64->256 conv: 5.4 -> 3.8 us per workgroup). Same pass structure and barrier count as
// igemm_epilogue.
template <typename T, class CFG>
DEV void igemm_epilogue_plain(f32x16 (&acc)[CFG::RM][CFG::RN], const Epilogue& ep, const RowMap& rm, char* smem, int M, int N, int m0, int n0,
                              int tid, int lane, int wave, int wm0, int wn0) {
  constexpr int BN = CFG::BN;
  constexpr int RM = CFG::RM, RN = CFG::RN;
  constexpr int CPRE = BN / 8, RPSE = 256 / CPRE, ITER = CFG::WM / RPSE;
  const int ecol = (tid % CPRE) * 8, erow0 = tid / CPRE;
  const int gcol = n0 + ecol;
  const bool colok = gcol < N;
  const bool stats = ep.colsum != nullptr;
  float csum[8], csq[8], bias[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { csum[e] = 0.f; csq[e] = 0.f; bias[e] = (colok && ep.bias) ? ep.bias[gcol + e] : 0.f; }
  for (int pass = 0; pass < CFG::WAVES_M; ++pass) {
    if (wave / CFG::WAVES_N == pass) {
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            int col = wn0 + j * 32 + (lane & 31);
            *(float*)(smem + row * CFG::EPI_PITCH + col * 4) = acc[i][j][r];
          }
    }
    lds_barrier();
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int rr = erow0 + it * RPSE;
      const int grow = m0 + pass * CFG::WM + rr;
      if (colok && grow < M) {
        const float* src = (const float*)(smem + rr * CFG::EPI_PITCH + ecol * 4);
        f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * ep.alpha + bias[e];
        store8((bf16*)ep.out + (map_row(rm, grow) * ep.ldc + gcol), v);
        if (stats) {
          round8_bf16(v);   // statistics of what was stored
#pragma unroll
          for (int e = 0; e < 8; ++e) { csum[e] += v[e]; csq[e] += v[e] * v[e]; }
        }
      }
    }
    lds_barrier();
  }
  if (stats) {
    float* crep = ep.colsum + (ep.colsum_replicas > 1 ? (size_t)(blockIdx.x % ep.colsum_replicas) * ep.colsum_stride : 0);
    float* red = (float*)smem;                      // [RPSE][CPRE*16]
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[erow0 * (CPRE * 16) + (tid % CPRE) * 16 + e] = csum[e];
      red[erow0 * (CPRE * 16) + (tid % CPRE) * 16 + 8 + e] = csq[e];
    }
    lds_barrier();
    for (int idx = tid; idx < CPRE * 16; idx += 256) {
      float sacc = 0.f;
      for (int r = 0; r < RPSE; ++r) sacc += red[r * (CPRE * 16) + idx];
      int chunk = idx / 16, e = idx % 16;
      int col = n0 + chunk * 8 + (e & 7);
      if (col < N && (e < 8 || ep.colsum_rows != 1)) atomic_add_f32(crep + (e >= 8 ? N : 0) + col, sacc);
    }
  }
}

// BatchNorm-backward epilogue (conv dgrad inside the ResNet backward; see HEAVY above) in its own layout: the kernel runs at two
// workgroups per CU anyway (registers), so the whole BM x BN accumulator tile is staged in LDS at once (one barrier instead of two per
// wave row) and every thread then walks its BM/RPSE rows with the three extra operands (mask source, BatchNorm input, residual) of the
// next rows already in flight: their latency is paid once per tile, not once per row.
// What one workgroup of igemm_dma_bn_kernel keeps across ALL the tiles it walks (they share one column tile): the BatchNorm means of the
// thread's 8 columns — 8 x replicas dependent L2 loads, a latency chain that used to open every tile's epilogue — and the running column
// sums, folded through LDS and added with float atomics ONCE per workgroup instead of once per tile. (An ablation build without any global
// memory traffic still took 42 of the 85 us of the 1024 <- 256 dgrad at 14 x 14: the per-tile fixed costs, not the bytes, were half the launch.)
struct BnEpiState {
  float bn_mean[8], csum[8], csq[8];
  float bias[8];          // FORM 5 only (clite_epilogue.bias in the BatchNorm-backward form: the folded BatchNorm backward's constant row)
};
template <class CFG>
DEV void bn_epi_begin(BnEpiState& st, const Epilogue& ep, int N, int n0, int tid) {
  constexpr int CPRE = CFG::BN / 8;
  const int gcol = n0 + (tid % CPRE) * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) { st.bn_mean[e] = 0.f; st.csum[e] = 0.f; st.csq[e] = 0.f; st.bias[e] = (gcol < N && ep.bias) ? ep.bias[gcol + e] : 0.f; }
  if (gcol < N && ep.bn_y) {
    for (int r = 0; r < ep.bn_replicas; ++r)
#pragma unroll
      for (int e = 0; e < 8; ++e) st.bn_mean[e] += ep.bn_stats[(size_t)r * ep.bn_rstride + gcol + e];
#pragma unroll
    for (int e = 0; e < 8; ++e) st.bn_mean[e] *= ep.bn_inv_count;
  }
}
// fold the workgroup's column sums through LDS (free at this point) and add them: one atomic per column and statistic
template <class CFG>
DEV void bn_epi_finish(const BnEpiState& st, const Epilogue& ep, char* smem, int N, int n0, int tid) {
  constexpr int CPRE = CFG::BN / 8, RPSE = 256 / CPRE;
  if (!ep.colsum) return;
  const int erow0 = tid / CPRE;
  float* crep = ep.colsum + (ep.colsum_replicas > 1 ? (size_t)(blockIdx.x % ep.colsum_replicas) * ep.colsum_stride : 0);
  float* red = (float*)smem;                      // [RPSE][CPRE*16]
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[erow0 * (CPRE * 16) + (tid % CPRE) * 16 + e] = st.csum[e];
    red[erow0 * (CPRE * 16) + (tid % CPRE) * 16 + 8 + e] = st.csq[e];
  }
  lds_barrier();
  for (int idx = tid; idx < CPRE * 16; idx += 256) {
    float sacc = 0.f;
    for (int r = 0; r < RPSE; ++r) sacc += red[r * (CPRE * 16) + idx];
    int chunk = idx / 16, e = idx % 16;
    int col = n0 + chunk * 8 + (e & 7);
    if (col < N && (e < 8 || ep.colsum_rows != 1)) atomic_add_f32(crep + (e >= 8 ? N : 0) + col, sacc);
  }
}

// The epilogue operands of one thread's BM / RPSE rows (relu' bits, BatchNorm input, residual, tensor-form mask), in registers. A struct of
// its own so that igemm_dma_bn_kernel can request the first rows of a tile BEFORE the tile's main loop: the requests need only addresses,
// not accumulators, and `vmcnt` retires in issue order — the wait on the first operand tile covers them, so their HBM round trip runs
// beside the ring fill instead of opening the epilogue (one of a short tile's three serial round trips: ring fill, rows 0..AHEAD-1, the rest).
template <typename T, class CFG, int FORM>
struct BnRows {
  static constexpr int CPRE = CFG::BN / 8, RPSE = 256 / CPRE, ROWS_PT = CFG::BM / RPSE;
  Raw8<T> pa[ROWS_PT], py[ROWS_PT], pr[ROWS_PT];
  uint32_t pb[ROWS_PT];          // packed relu' bits of the row's 8 columns (clite_epilogue.relu_bits): one byte instead of a 16-byte chunk of dact_aux
  uint32_t gix[ROWS_PT];
  bool okr[ROWS_PT];
  // Every operand load is BRANCH-FREE: a buffer load whose offset is OOB_OFF (-> zeros, no memory traffic) when the operand is absent or
  // the row / column is out of range. With the loads inside `if (ep.bn_y) ...` / `if (okr) ...` branches the compiler's waitcnt insertion
  // put an `s_waitcnt vmcnt(0)` behind the first load of every request (found in the ISA, round 3): the rows "in flight ahead" were in fact
  // fetched one at a time, and the epilogue phase alone ran at 2.2 TB/s (ablation without the operand DMA: 71 of the 85 us of the
  // 1024 <- 256 dgrad at 14 x 14).
  DEV void request(int q, const Epilogue& ep, const RowMap& rm, int M, int N, int m0, int n0, int tid) {
    const int gcol = n0 + (tid % CPRE) * 8;
    // (FORM 5 = FORM 1 + a bias row: the same operands)
    const bool has_bits = FORM ? FORM != 4 : ep.relu_bits != nullptr, has_aux = FORM ? false : ep.dact_aux != nullptr, has_y = FORM ? FORM != 4 : ep.bn_y != nullptr,
               has_res = FORM ? (FORM == 2 || FORM == 3) : ep.residual != nullptr;
    const rsrc_t r_bits = make_rsrc(ep.relu_bits, RSRC_WHOLE), r_aux = make_rsrc(ep.dact_aux, RSRC_WHOLE), r_y = make_rsrc(ep.bn_y, RSRC_WHOLE),
                 r_res = make_rsrc(ep.residual, RSRC_WHOLE);
    const int grow = m0 + tid / CPRE + q * RPSE;
    okr[q] = gcol < N && grow < M;
    gix[q] = okr[q] ? (uint32_t)(map_row(rm, grow) * ep.ldc + gcol) : 0u;
    if constexpr (FORM == 4) return;          // the plain forward form has no epilogue operands: only the row's offset
#ifndef CLITE_EPI_ABLATE
#define CLITE_EPI_ABLATE 0       // diagnostic builds only: 1 = epilogue without its operand loads, 2 = without its stores
#endif
    const uint32_t eoff = (CLITE_EPI_ABLATE & 1) ? OOB_OFF : gix[q] * (uint32_t)sizeof(T);
    pb[q] = buf_load1(r_bits, (okr[q] && has_bits && !(CLITE_EPI_ABLATE & 1)) ? (gix[q] >> 3) : OOB_OFF);
    if constexpr (FORM == 0) pa[q].ldb(r_aux, (okr[q] && has_aux && !has_bits) ? eoff : OOB_OFF);
    py[q].ldb(r_y, (okr[q] && has_y) ? eoff : OOB_OFF);
    if constexpr (FORM == 3) {          // residual = the compact gradient of the stride-2 shortcut: present on every second pixel only
      const int rr = subsampled_row(rm, grow);
      pr[q].ldb(r_res, (okr[q] && rr >= 0 && !(CLITE_EPI_ABLATE & 1)) ? (uint32_t)(rr * ep.ldc + gcol) * (uint32_t)sizeof(T) : OOB_OFF);
    } else if constexpr (FORM == 0) {
      uint32_t roff = eoff;
      if (rm.on == 2) {
        const int rr = subsampled_row(rm, grow);
        roff = rr >= 0 ? (uint32_t)(rr * ep.ldc + gcol) * (uint32_t)sizeof(T) : OOB_OFF;
      }
      pr[q].ldb(r_res, (okr[q] && has_res) ? roff : OOB_OFF);
    } else if constexpr (FORM == 2) {
      pr[q].ldb(r_res, okr[q] ? eoff : OOB_OFF);
    }
  }
};

// (FORM 3, round 4 = FORM 2 with the residual read through RowMap's subsampled-row map: the block-input gradient of a stride-2 downsample block.)
// FORM: what the launch needs, fixed at compile time. 0 = every combination clite_epilogue allows in this form, decided by run-time flags and
// selects (tests, f32, the tensor form of the mask). 1 / 2 = the two forms the bf16 ResNet backward launches — packed relu' bits, BatchNorm input,
// bf16 output, alpha = 1; 2 adds the residual with the mask applied after it (the block-input gradient). The ISA of the run-time form had ~400
// instructions per row of 8 elements (flag selects, both store paths, 64-bit index arithmetic) and an ablation build without ANY global memory
// traffic still took half the launch: the epilogue was instruction-bound, not byte-bound. The specialised forms issue ~1/3 of that.
// NREQ: the thread's first NREQ rows (processing order) were requested by the caller (BnRows::request, before the main loop).
template <typename T, class CFG, int FORM = 0, int NREQ = 0>
DEV void igemm_epilogue_bn(f32x16 (&acc)[CFG::RM][CFG::RN], BnEpiState& st, BnRows<T, CFG, FORM>& rows, const Epilogue& ep, const RowMap& rm, char* smem, int M, int N,
                           int m0, int n0, int tid, int lane, int wave, int wm0, int wn0) {
  constexpr int BM = CFG::BM, BN = CFG::BN;
  constexpr int RM = CFG::RM, RN = CFG::RN;
  constexpr int CPRE = BN / 8, RPSE = 256 / CPRE, ROWS_PT = BM / RPSE;
  const int ecol = (tid % CPRE) * 8, erow0 = tid / CPRE;
  const int gcol = n0 + ecol;
  const bool colok = gcol < N;
  // The accumulator tile goes through LDS in NP passes of BM / NP rows (NP = 1: the whole tile at once, 68 KB = two workgroups per CU; NP = 2 in
  // the specialised forms: 34 KB, under the 48 KB operand ring, so THREE workgroups fit a CU). The ablations of round 3 showed the launch's
  // parts — compute, operand DMA, epilogue loads, epilogue stores — adding up instead of overlapping: with two waves per SIMD both are
  // usually parked on memory at the same time; a third resident workgroup is what hides that.
  // Pass p takes the p-th 32-row MFMA block of EVERY wave (not the first half of the waves), so that each wave's accumulators die by halves
  // and no wave carries all 64 of them through the VALU-heavy row loop of the other pass: tile row R belongs to pass (R / 32) % RM and sits at
  // image row (R / 32 / RM) * 32 + R % 32.
  constexpr int NP = (FORM && CLITE_BN_HALF) ? RM : 1, QPP = ROWS_PT / NP;
  static_assert(NP == 1 || (RM == 2 && ROWS_PT % NP == 0 && 32 % RPSE == 0), "pass structure");
  auto stage = [&](int p) {
#pragma unroll
    for (int i = 0; i < RM; ++i) {
      if (NP > 1 && i != p) continue;
#pragma unroll
      for (int j = 0; j < RN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int row = (NP > 1 ? (wm0 / CFG::WM) * 32 : wm0 + i * 32) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          int col = wn0 + j * 32 + (lane & 31);
          *(float*)(smem + row * CFG::EPI_PITCH + col * 4) = acc[i][j][r];
        }
    }
  };
  // processing order of the thread's ROWS_PT rows: pass 0's rows first. Row index q (tile row erow0 + q * RPSE) -> pass ((q * RPSE) / 32) % NP
  auto row_of = [](int k) {          // k-th row in processing order
    if (NP == 1) return k;
    int p = k / QPP, n = k % QPP, seen = 0;
    for (int q = 0; q < ROWS_PT; ++q)
      if (((q * RPSE) / 32) % NP == p) { if (seen == n) return q; ++seen; }
    return 0;
  };
  stage(0);
  float (&bn_mean)[8] = st.bn_mean;
  float (&csum)[8] = st.csum;
  float (&csq)[8] = st.csq;
  // Rows whose operands are in flight ahead of the row being written. Measured alternatives (round 2, MI355X, in-step per-launch times):
  //  * all ROWS_PT rows ahead (the registers are there: the whole-tile staging holds the kernel at two workgroups per CU anyway):
  //    no faster — 1024 -> 256 1x1 dgrad 80.7 vs 82.5 us, 256 -> 64 191 vs 217;
  //  * staging one wave row at a time (34 KB, under the operand ring) + __launch_bounds__(256, 3) for three workgroups per CU: the
  //    allocator then spills 20 registers and the launches get slower — 256 -> 64 191 -> 267 us, 64 -> 256 97 -> 116, step 18.7 -> 19.2 ms.

  constexpr int AHEAD = (FORM && CLITE_BN_HALF) ? 2 : (FORM ? CLITE_BN_AHEAD_FORM : CLITE_BN_AHEAD);
  static_assert(NREQ == 0 || NP == 1, "early requests assume the whole-tile staging order");
  const bool has_bits = FORM ? FORM != 4 : ep.relu_bits != nullptr, has_aux = FORM ? false : ep.dact_aux != nullptr, has_y = FORM ? FORM != 4 : ep.bn_y != nullptr;
  const bool mask_after = FORM ? (FORM == 2 || FORM == 3) : ep.mask_after_residual != 0;
  const bool to_f32 = FORM ? false : (ep.out_f32 || sizeof(T) == 4);
  const rsrc_t r_out = make_rsrc(ep.out, RSRC_WHOLE);
  Raw8<T> (&pa)[ROWS_PT] = rows.pa, (&py)[ROWS_PT] = rows.py, (&pr)[ROWS_PT] = rows.pr;
  uint32_t (&pb)[ROWS_PT] = rows.pb, (&gix)[ROWS_PT] = rows.gix;
  bool (&okr)[ROWS_PT] = rows.okr;
  auto request = [&](int q) { rows.request(q, ep, rm, M, N, m0, n0, tid); };
#pragma unroll
  for (int k = NREQ; k < AHEAD && k < ROWS_PT; ++k) request(row_of(k));
  lds_barrier();
  EPI_STAMP(6);
  BN_EPI_PHASE(4);        // accumulators staged, first requests issued
#pragma unroll
  for (int k = 0; k < ROWS_PT; ++k) {
    const int q = row_of(k);
    if (NP > 1 && k > 0 && k % QPP == 0) {          // next pass: every thread is past its reads of the previous image, then every wave stages
      lds_barrier();
      stage(k / QPP);
      lds_barrier();
    }
    if (k + AHEAD < ROWS_PT && k + AHEAD >= NREQ) request(row_of(k + AHEAD));
    const int irow = NP == 1 ? erow0 + q * RPSE : ((erow0 + q * RPSE) / 32 / RM) * 32 + (erow0 + q * RPSE) % 32;
    const float* src = (const float*)(smem + irow * CFG::EPI_PITCH + ecol * 4);
    f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 4);
    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    float msk[8], av[8], rv[8], yv[8];
    if constexpr (FORM != 4) py[q].get(yv);
    if constexpr (FORM == 4) {
      // FORM 4 (round 4): the plain FORWARD epilogue of the HBM-bound 1 x 1 convolutions in this kernel's row-range persistent structure - bf16 store
      // of the accumulators, column sums (sum v, sum v^2) of the stored values; no operands, no mask (out-of-range rows have zero accumulators)
#pragma unroll
      for (int e = 0; e < 8; ++e) yv[e] = 0.f;
    } else if constexpr (FORM == 0) {
      pa[q].get(av);
      pr[q].get(rv);          // zeros when there is no residual (or the row is out of range)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        // relu' mask: packed bits, or the sign of the tensor operand, or none (ReLU' only here: check_ep enforces dact == 1)
        msk[e] = has_bits ? ((pb[q] >> e) & 1u ? 1.f : 0.f) : (has_aux ? (av[e] > 0.f ? 1.f : 0.f) : 1.f);
        const float a = v[e] * ep.alpha + st.bias[e];
        v[e] = mask_after ? (a + rv[e]) * msk[e] : a * msk[e] + rv[e];
        if (!okr[q]) v[e] = 0.f;          // (out-of-range rows / columns: nothing stored, nothing added to the statistics)
      }
    } else {
      // out-of-range rows / columns need no select: their accumulators are zero (operand rows / columns past the range gather as zeros) and
      // every operand load returned zeros, so v = 0 and nothing is added to the statistics; the store offset is OOB_OFF
      if constexpr (FORM == 2 || FORM == 3) {
        pr[q].get(rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (pb[q] >> e) & 1u ? v[e] + rv[e] : 0.f;
      } else if constexpr (FORM == 5) {
        // the folded BatchNorm backward (clite_conv_dgrad_bnfold): + the constant row, then the mask. Out-of-range rows / columns read zero bits
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (pb[q] >> e) & 1u ? v[e] + st.bias[e] : 0.f;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (pb[q] >> e) & 1u ? v[e] : 0.f;
      }
    }
    if (to_f32) {
      union { f32x4 f; u32x4 u; } lo, hi;
      lo.f = f32x4{v[0], v[1], v[2], v[3]}; hi.f = f32x4{v[4], v[5], v[6], v[7]};
      const uint32_t o = (okr[q] && !(CLITE_EPI_ABLATE & 2)) ? gix[q] * 4u : OOB_OFF;
      buf_store16(r_out, o, lo.u);
      buf_store16(r_out, o == OOB_OFF ? OOB_OFF : o + 16, hi.u);
    } else {
      Chunk16 ch;
#pragma unroll
      for (int e = 0; e < 8; ++e) ch.e[e] = f2bf(v[e]);
      buf_store16(r_out, (okr[q] && !(CLITE_EPI_ABLATE & 2)) ? gix[q] * 2u : OOB_OFF, ch.u);
      round8_bf16(v);   // statistics of what was stored
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { csum[e] += v[e]; csq[e] += has_y ? v[e] * (yv[e] - bn_mean[e]) : v[e] * v[e]; }
  }
  lds_barrier();
  EPI_STAMP(7);
  BN_EPI_PHASE(5);        // row loop: operands in, results out, statistics
}

#if defined(CLITE_DIAG) && CLITE_DIAG
// Round-1 register-staged engine (global -> VGPR -> ds_write_b128, two LDS stages): diagnostic builds only, as the A/B yardstick for the
// LDS-DMA kernels of igemm_dma.h. The product library does not contain it.
template <typename T, class CFG, class LA, class LB>
__global__ __launch_bounds__(256) void igemm_kernel(LA la, LB lb, Epilogue ep, RowMap rm, int M, int N, int ktiles, int ktiles_per_split) {
  constexpr int BM = CFG::BM, BN = CFG::BN, BK = CFG::BK;
  constexpr int RM = CFG::RM, RN = CFG::RN;
  constexpr int STAGE = LA::BYTES + LB::BYTES;
  constexpr int SMEM = (2 * STAGE > CFG::EPI_BYTES) ? 2 * STAGE : CFG::EPI_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm0 = (wave / CFG::WAVES_N) * CFG::WM;
  const int wn0 = (wave % CFG::WAVES_N) * CFG::WN;

  // XCD-aware tile order: workgroups b and b+8 share an XCD (and its 4 MiB L2) under round-robin dispatch, so give
  // each XCD a contiguous run of logical tiles (bijective for any grid size); placement affects speed only.
  const int nwg = gridDim.x, xcd = blockIdx.x & 7, xq = nwg >> 3, xr = nwg & 7;
  const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
  const int tiles_n = (N + BN - 1) / BN;
  const int tm = wg / tiles_n;
  const int tn = wg - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int t_begin = blockIdx.z * ktiles_per_split;
  int t_end = t_begin + ktiles_per_split;
  if (t_end > ktiles) t_end = ktiles;

  typename LA::State sa;
  typename LB::State sb;
  la.init(sa, m0, tid, t_begin);
  lb.init(sb, n0, tid, t_begin);

  f32x16 acc[RM][RN];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4 ra[LA::NCH], rb[LB::NCH];
  if (t_begin < t_end) {
    la.load(sa, ra);
    lb.load(sb, rb);
    LA::store(smem, tid, ra);
    LB::store(smem + LA::BYTES, tid, rb);
  }
  __syncthreads();

  for (int t = t_begin; t < t_end; ++t) {
    const int cur = (t - t_begin) & 1;
    const bool more = t + 1 < t_end;
    if (more) {
      la.load(sa, ra);
      lb.load(sb, rb);
    }
    const char* abuf = smem + cur * STAGE;
    const char* bbuf = abuf + LA::BYTES;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        bf16x8 af[RM], bfr[RN];
#pragma unroll
        for (int i = 0; i < RM; ++i) af[i] = LA::frag(abuf, wm0 + i * 32, ks, lane);
#pragma unroll
        for (int j = 0; j < RN; ++j) bfr[j] = LB::frag(bbuf, wn0 + j * 32, ks, lane);
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
          for (int j = 0; j < RN; ++j) acc[i][j] = mfma32_bf16(af[i], bfr[j], acc[i][j]);
      }
    } else {   // exact-f32 parity mode: v_mfma_f32_32x32x2_f32, a k-ordered fmaf chain per output
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        float af[RM], bfr[RN];
#pragma unroll
        for (int i = 0; i < RM; ++i) af[i] = LA::frag32(abuf, wm0 + i * 32, kk, lane);
#pragma unroll
        for (int j = 0; j < RN; ++j) bfr[j] = LB::frag32(bbuf, wn0 + j * 32, kk, lane);
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
          for (int j = 0; j < RN; ++j) acc[i][j] = mfma32_f32(af[i], bfr[j], acc[i][j]);
      }
    }
    if (more) {
      char* nbuf = smem + (cur ^ 1) * STAGE;
      LA::store(nbuf, tid, ra);
      LB::store(nbuf + LA::BYTES, tid, rb);
    }
    __syncthreads();
  }

  igemm_epilogue<T, CFG>(acc, ep, rm, smem, M, N, m0, n0, tid, lane, wave, wm0, wn0);
}

#endif  // CLITE_DIAG

}  // namespace clite
#endif  // CLITE_IGEMM_H
