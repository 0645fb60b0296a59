// LDS-DMA pipelined variant of the implicit-GEMM engine (same math, operands, epilogue and C ABI as igemm.h).
//
// Operand tiles go global -> LDS with `buffer_load_dwordx4 ... lds` (no VGPR staging, no ds_write), three LDS stages, one raw
// s_barrier per K tile and a counted `s_waitcnt vmcnt(N)` that leaves the next tile's loads in flight across the barrier:
//
//   prologue: issue(t0 -> buf0), issue(t0+1 -> buf1)        (NSTAGE = 3; NSTAGE = 4 keeps one more tile in flight for small grids)
//   for t:    vmcnt(loads of one tile)   // tile t landed (this wave's part) ...
//             s_barrier                  // ... and everyone's part; also: everyone finished computing tile t-1
//             issue(t+2 -> buf (t+2)%3)  // overwrites the buffer of tile t-1
//             compute(buf t%3)
//
// The DMA destination is lane-linear (wave-uniform base + lane*16 B), so rows cannot be padded; bank conflicts are removed by
// an XOR swizzle applied to the SOURCE chunk each lane fetches and to the fragment read address (the same involution on both
// sides):
//   KC image [rows][64 B]:   16-B chunk c of row r sits in slot c ^ ((r>>2)&3)          -> conflict-free ds_read_b128
//   XC image [BK][cols*sizeof(T)]: 64-B segment s of k-row k sits in segment s ^ f(k)   -> conflict-free ds_read_b64_tr_b16
#ifndef CLITE_IGEMM_DMA_H
#define CLITE_IGEMM_DMA_H
#ifndef CLITE_ABLATE
#define CLITE_ABLATE 0      // diagnostic builds only (tools/ablate.sh): 1 = main loop without LDS reads / MFMAs, 2 = without the DMA loads
#endif
#ifndef CLITE_STAMP
#define CLITE_STAMP 0       // diagnostic builds only: per-workgroup phase timestamps (s_memrealtime, 100 MHz) into clite_dbg[]
#endif
#if CLITE_STAMP
__device__ unsigned long long clite_dbg[8 * 65536];
#define EPI_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 65536) clite_dbg[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 65536) clite_dbg[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// the row-range persistent kernel walks several tiles: it ACCUMULATES phase durations (thread 0's clock) in LDS and writes the sums at the end
__shared__ unsigned long long clite_ph[8];
#define PHASE(i) do { if (threadIdx.x == 0) { unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); clite_ph[i] += n_ - clite_ph[7]; clite_ph[7] = n_; } } while (0)
#define BN_EPI_PHASE(i) PHASE(i)
#else
#define STAMP(i) do {} while (0)
#define PHASE(i) do {} while (0)
#endif
#include "igemm.h"

namespace clite {

// ---- KC gather (activations forward, dY in dgrad, weights forward). One wave instruction = 16 rows x 64 B.
template <typename T, int ROWS, int BK, bool DGRAD>
struct DmaKC {
  static_assert(BK * sizeof(T) == 64, "a K tile row is one 64-byte line");
  static constexpr int EPC = 16 / (int)sizeof(T);
  static constexpr int NI = ROWS / 64;               // DMA instructions per wave per tile (4 waves x 16 rows each)
  static constexpr int BYTES = ROWS * 64;
  const void* ptr;
  uint32_t bytes;
  ConvGeom g;

  struct State {
    rsrc_t rs;
    int base[NI], h0[NI], w0[NI];
    uint32_t off[NI];           // byte offset of this lane's chunk for the current K tile (valid when ok[j])
    bool ok[NI];                // window position (r, s) lands inside the tensor for row j
    int kc;                     // element offset inside the K tile of the chunk this lane fetches (swizzled)
    int r, s, c0;
  };
  // offsets of all rows for the current (r, s, c0): done once per window position, then advanced by 64 B per K tile
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
      st.off[j] = (uint32_t)(st.base[j] + hi * g.sH + wi * g.sW + st.c0 + st.kc) * (uint32_t)sizeof(T);
    }
  }
  DEV void init(State& st, int row0, int wave, int lane, int t_begin) const {
    st.rs = make_rsrc(ptr, bytes);
    st.kc = ((lane & 3) ^ ((lane >> 4) & 3)) * EPC;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      int row = row0 + (wave * NI + j) * 16 + (lane >> 2);
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
    int k0 = t_begin * BK;
    int rs = k0 / g.C;
    st.c0 = k0 - rs * g.C;
    st.r = rs / g.S;
    st.s = rs - st.r * g.S;
    locate(st);
  }
  DEV void issue(State& st, char* lds, int wave) const {
    bool kvalid = st.c0 + st.kc < g.C;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      uint32_t off = (kvalid && st.ok[j]) ? st.off[j] : OOB_OFF;
      buf_load16_lds(st.rs, off, lds + (wave * NI + j) * 1024);
      st.off[j] += 64;
    }
    st.c0 += BK;
    if (st.c0 >= g.C) {          // next window position (block-uniform branch)
      st.c0 = 0;
      if (++st.s == g.S) { st.s = 0; ++st.r; }
      locate(st);
    }
  }
  // byte offset (inside the operand image) of the fragment of the 32-row block at x0, k-step ks: loop-invariant per lane
  DEV static int frag_off(int x0, int ks, int lane) {
    int r = x0 + (lane & 31), c = ks * 2 + (lane >> 5);
    return r * 64 + ((c ^ ((r >> 2) & 3)) << 4);
  }
  DEV static bf16x8 frag_at(const char* p) {
    Chunk16 ch;
    ch.u = *(const u32x4*)p;
    return ch.h;
  }
  DEV static float frag32(const char* lds, int x0, int kk, int lane) {
    int r = x0 + (lane & 31), k = kk * 2 + (lane >> 5);           // k in [0,16)
    return *(const float*)(lds + r * 64 + (((k >> 2) ^ ((r >> 2) & 3)) << 4) + (k & 3) * 4);
  }
  // f32 only (a K tile = 16 values = one bf16 32 x 32 x 16 k-step): the 8 consecutive k = 8 (lane >> 5) .. + 7 of row x0 + (lane & 31) — the operand
  // of one MFMA of the split-bf16 form (split_bf16x3): two 16-byte reads
  DEV static void frag8_f32(const char* lds, int x0, int lane, float (&v)[8]) {
    const int r = x0 + (lane & 31), h = lane >> 5, sw = (r >> 2) & 3;
    const f32x4 a = *(const f32x4*)(lds + r * 64 + (((2 * h) ^ sw) << 4)), b = *(const f32x4*)(lds + r * 64 + (((2 * h + 1) ^ sw) << 4));
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
  }
};

// 64-byte-segment swizzle of an XC image row k (row = COLS*sizeof(T) bytes)
template <int ROWBYTES> DEV int xc_seg_xor(int k) {
  return ROWBYTES == 128 ? ((k >> 1) & 1) : (k & 3);      // 128 B rows: 2 k-rows per 256-B bank row; >= 256 B rows: one or less
}

// ---- XC strided (weights in dgrad, dY in wgrad, linear weights in input-grad): element (k, x) at ptr[(k0+krow)*ld + rs*Cx + x]
template <typename T, int COLS, int BK, int NW = 4>           // NW: waves per workgroup that share the tile's DMA instructions
struct DmaXCStrided {
  static constexpr int EPC = 16 / (int)sizeof(T);
  static constexpr int ROWB = COLS * (int)sizeof(T);         // bytes per k-row of the image
  static constexpr int CPR = ROWB / 16;                      // 16-B chunks per k-row
  static constexpr int RPI = 64 / CPR;                       // k-rows per wave instruction
  static constexpr int NI = BK / (NW * RPI);                 // instructions per wave per tile
  static constexpr int BYTES = BK * ROWB;
  static_assert(CPR <= 64 && 64 % CPR == 0 && BK % (NW * RPI) == 0 && NI >= 1, "tile shape");
  const void* ptr;
  uint32_t bytes;
  int ld, Cx, Ck, RS;

  struct State {
    rsrc_t rs_;
    int x[NI];        // first element of the (swizzled) source chunk this lane fetches in instruction j, or -1 if out of range
    int krow[NI];     // k-row inside the tile
    uint32_t off[NI]; // byte offset for the current K tile
    int rs, k0;
  };
  DEV void locate(State& st) const {
#pragma unroll
    for (int j = 0; j < NI; ++j)
      st.off[j] = (uint32_t)((st.k0 + st.krow[j]) * ld + st.rs * Cx + (st.x[j] < 0 ? 0 : st.x[j])) * (uint32_t)sizeof(T);
  }
  DEV void init(State& st, int x0, int wave, int lane, int t_begin) const {
    st.rs_ = make_rsrc(ptr, bytes);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      int krow = (wave * NI + j) * RPI + lane / CPR;
      int c = (lane % CPR) ^ (xc_seg_xor<ROWB>(krow) << 2);   // source chunk that lands in this lane's slot
      int x = x0 + c * EPC;
      st.krow[j] = krow;
      st.x[j] = x < Cx ? x : -1;
    }
    int kk = t_begin * BK;
    st.rs = (RS == 1) ? 0 : kk / Ck;
    st.k0 = kk - st.rs * Ck;
    locate(st);
  }
  DEV void issue(State& st, char* lds, int wave, int, int) const {
    const uint32_t step = (uint32_t)(BK * ld) * (uint32_t)sizeof(T);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      bool v = st.x[j] >= 0 && st.k0 + st.krow[j] < Ck && st.rs < RS;
      buf_load16_lds(st.rs_, v ? st.off[j] : OOB_OFF, lds + (wave * NI + j) * 1024);
      st.off[j] += step;
    }
    st.k0 += BK;
    if (st.k0 >= Ck && RS > 1) { st.k0 = 0; ++st.rs; locate(st); }
  }
  // k and k+4 get the same segment swizzle for every supported row size, so the second transposed read is +4 rows
  DEV static int frag_off(int x0, int ks, int lane) {
    int x = x0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);     // 4 consecutive x of one 16-B chunk half
    int k = ks * 16 + 8 * (lane >> 5) + ((lane >> 2) & 3);
    int c = (x * 2) >> 4, within = (x * 2) & 15;
    return k * ROWB + ((c ^ (xc_seg_xor<ROWB>(k) << 2)) << 4) + within;
  }
  DEV static bf16x8 frag_at(const char* p) {
    s16x4 lo = lds_read_tr16(p);
    s16x4 hi = lds_read_tr16(p + 4 * ROWB);
    union { s16x4 v[2]; bf16x8 h; } u;
    u.v[0] = lo; u.v[1] = hi;
    return u.h;
  }
  DEV static float frag32(const char* lds, int x0, int kk, int lane) {
    int k = kk * 2 + (lane >> 5), x = x0 + (lane & 31);
    int c = (x * 4) >> 4, within = (x * 4) & 15;
    return *(const float*)(lds + k * ROWB + ((c ^ (xc_seg_xor<ROWB>(k) << 2)) << 4) + within);
  }
  // (f32, split-bf16 form) the 8 k-rows 8 (lane >> 5) .. + 7 of column x0 + (lane & 31): eight 4-byte reads — the contraction index is the slow one
  DEV static void frag8_f32(const char* lds, int x0, int lane, float (&v)[8]) {
    const int x = x0 + (lane & 31), c = (x * 4) >> 4, within = (x * 4) & 15, k0 = 8 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *(const float*)(lds + (k0 + j) * ROWB + ((c ^ (xc_seg_xor<ROWB>(k0 + j) << 2)) << 4) + within);
  }
};

// ---- XC gather (wgrad's activation operand): k = output pixel p, x = (r, s, ci)
template <typename T, int COLS, int BK, int NW = 4>
struct DmaXCGather {
  typedef DmaXCStrided<T, COLS, BK, NW> L;
  static constexpr int EPC = L::EPC, ROWB = L::ROWB, CPR = L::CPR, RPI = L::RPI, NI = L::NI, BYTES = L::BYTES;
  const void* ptr;
  uint32_t bytes;
  ConvGeom g;

  struct State {
    rsrc_t rs_;
    int xoff[NI];     // r*sH + s*sW + ci of this lane's source chunk in instruction j, or -1 if the column is out of range
    int r[NI], s[NI], krow[NI];
    int p0;
  };
  DEV void init(State& st, int x0, int wave, int lane, int t_begin) const {
    st.rs_ = make_rsrc(ptr, bytes);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      int krow = (wave * NI + j) * RPI + lane / CPR;
      int c = (lane % CPR) ^ (xc_seg_xor<ROWB>(krow) << 2);
      int x = x0 + c * EPC;
      int rs = x / g.C;
      int ci = x - rs * g.C;
      st.r[j] = rs / g.S;
      st.s[j] = rs - st.r[j] * g.S;
      st.krow[j] = krow;
      st.xoff[j] = x < g.R * g.S * g.C ? ci : -1;
    }
    st.p0 = t_begin * BK;
  }
  DEV void issue(State& st, char* lds, int wave, int, int) const {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      int p = st.p0 + st.krow[j];
      bool v = st.xoff[j] >= 0 && p < g.rows;
      uint32_t n = fd_div(p, g.div_hw);
      uint32_t rem = p - n * g.div_hw.d;
      uint32_t ho = fd_div(rem, g.div_w);
      uint32_t wo = rem - ho * g.div_w.d;
      int hi = (int)ho * g.stride - g.pad + st.r[j];
      int wi = (int)wo * g.stride - g.padw + st.s[j];
      v = v && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
      uint32_t off = v ? (uint32_t)((int)n * g.sN + hi * g.sH + wi * g.sW + st.xoff[j]) * (uint32_t)sizeof(T) : OOB_OFF;
      buf_load16_lds(st.rs_, off, lds + (wave * NI + j) * 1024);
    }
    st.p0 += BK;
  }
  DEV static int frag_off(int x0, int ks, int lane) { return L::frag_off(x0, ks, lane); }
  DEV static bf16x8 frag_at(const char* p) { return L::frag_at(p); }
  DEV static float frag32(const char* lds, int x0, int kk, int lane) { return L::frag32(lds, x0, kk, lane); }
  DEV static void frag8_f32(const char* lds, int x0, int lane, float (&v)[8]) { L::frag8_f32(lds, x0, lane, v); }
};

// uniform issue() signature over the three loaders
template <class L> struct DmaIssue {
  DEV static void go(const L& l, typename L::State& st, char* lds, int wave, int lane, int x0) { l.issue(st, lds, wave, lane, x0); }
};
template <typename T, int ROWS, int BK, bool D> struct DmaIssue<DmaKC<T, ROWS, BK, D>> {
  typedef DmaKC<T, ROWS, BK, D> L;
  DEV static void go(const L& l, typename L::State& st, char* lds, int wave, int, int) { l.issue(st, lds, wave); }
};

// Split-bf16 form of an f32 product (round 4): x = hi + lo + O(2^-17 |x|) with hi = bf16(x), lo = bf16(x - hi); a b ~ lo_a hi_b + hi_a lo_b + hi_a hi_b
// on three bf16 MFMAs (products of bf16 values are exact in f32, the accumulation is f32): relative error ~2^-17 per product instead of the
// bf16 path's 2^-9, at 3/16 of the f32 MFMA's matrix-pipe time per product. f32 storage is untouched. Selected at run time for the exact-f32
// mode's launches (clite_set_f32_split); the default f32 form stays the k-ordered fmaf chain of v_mfma_f32_32x32x2_f32.
DEV void split_bf16x3(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const bf16 h = f2bf(v[e]);
    hi[e] = h;
    lo[e] = f2bf(v[e] - bf2f(h));
  }
}
// one K tile (16 f32 = one bf16 k-step) of the split form: fragments from the f32 LDS images, split in registers, 3 MFMAs per block
template <class CFG, class LA, class LB>
DEV void mma_tile_f32_split(const char* abuf, const char* bbuf, int wm0, int wn0, int lane, f32x16 (&acc)[CFG::RM][CFG::RN]) {
  constexpr int RM = CFG::RM, RN = CFG::RN;
  bf16x8 ah[RM], al[RM], bh[RN], bl[RN];
#pragma unroll
  for (int i = 0; i < RM; ++i) {
    float v[8];
    LA::frag8_f32(abuf, wm0 + i * 32, lane, v);
    split_bf16x3(v, ah[i], al[i]);
  }
#pragma unroll
  for (int j = 0; j < RN; ++j) {
    float v[8];
    LB::frag8_f32(bbuf, wn0 + j * 32, lane, v);
    split_bf16x3(v, bh[j], bl[j]);
  }
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j) {          // small terms first
      acc[i][j] = mfma32_bf16(al[i], bh[j], acc[i][j]);
      acc[i][j] = mfma32_bf16(ah[i], bl[j], acc[i][j]);
      acc[i][j] = mfma32_bf16(ah[i], bh[j], acc[i][j]);
    }
}

// EPI: 0 = generic fused epilogue, 1 = BatchNorm-backward epilogue (igemm_epilogue_bn), 2 = plain bf16 store (+ bias, + column statistics)
template <typename T, class CFG, class LA, class LB, int NSTAGE, int EPI = 0, bool SPLIT = false>
__global__ __launch_bounds__(256) void igemm_dma_kernel(LA la, LB lb, Epilogue ep, RowMap rm, int M, int N, int ktiles, int ktiles_per_split, int xsplits) {
  static_assert(!SPLIT || (sizeof(T) == 4 && CFG::BK == 16), "the split-bf16 form is an f32 form: one K tile = one bf16 k-step");
  constexpr int BM = CFG::BM, BN = CFG::BN, BK = CFG::BK;
  constexpr int RM = CFG::RM, RN = CFG::RN;
  constexpr int STAGE = LA::BYTES + LB::BYTES;
  constexpr int EPIB = EPI == 1 ? CFG::BM * CFG::EPI_PITCH : CFG::EPI_BYTES;      // the BatchNorm-backward epilogue stages the whole tile
  constexpr int SMEM = (NSTAGE * STAGE > EPIB) ? NSTAGE * STAGE : EPIB;
  constexpr int LOADS_PER_TILE = LA::NI + LB::NI;            // per wave
  __shared__ __attribute__((aligned(1024))) char smem[SMEM];

  STAMP(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = wave_uniform(tid >> 6);
  const int wm0 = (wave / CFG::WAVES_N) * CFG::WM;
  const int wn0 = (wave % CFG::WAVES_N) * CFG::WN;

  const int tiles_n = (N + BN - 1) / BN;
  int wg, zsplit;
  if (xsplits > 0) {
    // split-K with >= 8 splits (weight gradients): all output tiles of one K range run on ONE XCD (workgroups b and b+8 share an XCD),
    // so each operand panel is fetched into one L2 instead of eight; XCD x takes the splits x, x+8, ...
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

  // loop-invariant LDS offsets of this lane's MFMA fragments (bf16 path)
  int aoff[RM][BK / 16 > 0 ? BK / 16 : 1], boff[RN][BK / 16 > 0 ? BK / 16 : 1];
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
#pragma unroll
      for (int i = 0; i < RM; ++i) aoff[i][ks] = LA::frag_off(wm0 + i * 32, ks, lane);
#pragma unroll
      for (int j = 0; j < RN; ++j) boff[j][ks] = LB::frag_off(wn0 + j * 32, ks, lane);
    }
  }

  STAMP(1);
  // prologue: NSTAGE-1 tiles in flight
#pragma unroll
  for (int pz = 0; pz < NSTAGE - 1; ++pz) {
    if (t_begin + pz < t_end) {
      DmaIssue<LA>::go(la, sa, smem + pz * STAGE, wave, lane, m0);
      DmaIssue<LB>::go(lb, sb, smem + pz * STAGE + LA::BYTES, wave, lane, n0);
    }
  }
  STAMP(2);
  int buf = 0;
  for (int t = t_begin; t < t_end; ++t) {
    // tile t has landed once at most min(NSTAGE-2, tiles after t) younger tiles are still outstanding
    const int after = t_end - 1 - t;
    if (NSTAGE >= 5 && after >= 3) wait_vmcnt<3 * LOADS_PER_TILE>();
    else if (NSTAGE >= 4 && after >= 2) wait_vmcnt<2 * LOADS_PER_TILE>();
    else if (after >= 1) wait_vmcnt<LOADS_PER_TILE>();
    else wait_vmcnt<0>();
    barrier_raw();
    if (t == t_begin) STAMP(3);
    const char* abuf = smem + buf * STAGE;
    const char* bbuf = abuf + LA::BYTES;
#if CLITE_ABLATE != 1
    if constexpr (sizeof(T) == 2) {
      // the first k-step's fragment reads go out BEFORE the next tile's DMA is issued: the DMA issue (address selects + 4 DMA
      // instructions per wave) then overlaps the LDS latency instead of preceding it
      bf16x8 af0[RM], bf0[RN];
#pragma unroll
      for (int i = 0; i < RM; ++i) af0[i] = LA::frag_at(abuf + aoff[i][0]);
#pragma unroll
      for (int j = 0; j < RN; ++j) bf0[j] = LB::frag_at(bbuf + boff[j][0]);
#if CLITE_ABLATE != 2
      if (t + NSTAGE - 1 < t_end) {
        int nb = buf + NSTAGE - 1; if (nb >= NSTAGE) nb -= NSTAGE;
        DmaIssue<LA>::go(la, sa, smem + nb * STAGE, wave, lane, m0);
        DmaIssue<LB>::go(lb, sb, smem + nb * STAGE + LA::BYTES, wave, lane, n0);
      }
#endif
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
    } else {
#if CLITE_ABLATE != 2
      if (t + NSTAGE - 1 < t_end) {
        int nb = buf + NSTAGE - 1; if (nb >= NSTAGE) nb -= NSTAGE;
        DmaIssue<LA>::go(la, sa, smem + nb * STAGE, wave, lane, m0);
        DmaIssue<LB>::go(lb, sb, smem + nb * STAGE + LA::BYTES, wave, lane, n0);
      }
#endif
      if constexpr (SPLIT) {
        mma_tile_f32_split<CFG, LA, LB>(abuf, bbuf, wm0, wn0, lane, acc);
      } else {
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
    }
#else
    asm volatile("" :: "v"(abuf), "v"(bbuf));
#endif
    if (++buf == NSTAGE) buf = 0;
  }
  barrier_raw();          // every wave is past its last fragment read before the epilogue reuses the LDS
  STAMP(4);
  if constexpr (EPI == 1) {          // (the BatchNorm-backward form is launched as igemm_dma_bn_kernel; kept for diagnostic builds)
    BnEpiState est;
    BnRows<T, CFG, 0> rows;
    bn_epi_begin<CFG>(est, ep, N, n0, tid);
    igemm_epilogue_bn<T, CFG>(acc, est, rows, ep, rm, smem, M, N, m0, n0, tid, lane, wave, wm0, wn0);
    bn_epi_finish<CFG>(est, ep, smem, N, n0, tid);
  } else if constexpr (EPI == 2 && sizeof(T) == 2) igemm_epilogue_plain<T, CFG>(acc, ep, rm, smem, M, N, m0, n0, tid, lane, wave, wm0, wn0);
  else igemm_epilogue<T, CFG, false>(acc, ep, rm, smem, M, N, m0, n0, tid, lane, wave, wm0, wn0);
  STAMP(5);
}


// ---- BatchNorm-backward conv dgrad as a ROW-RANGE PERSISTENT kernel ---------------------------------------------------------------
// The HBM-bound 1x1 dgrads of the ResNet backward (K = 64 ... 512: a handful of K tiles, then an epilogue that reads two or three more tensors
// and writes one) run at two workgroups per CU, i.e. 512 resident slots, and their tile counts are a little more than a multiple of that
// (1568 tiles of 128 x 128 for 1024 -> 256 at 14 x 14: 3.06 rounds, the fourth 94 % empty; profiles/r2_layers.txt). Here every workgroup
// owns `rows_per_wg` consecutive GEMM rows of ONE column tile and walks them in steps of BM; the last step is a partial tile (rows past the
// range read as zero / are not stored), which costs a short main loop and an epilogue proportional to its rows. With rows_per_wg =
// ceil(M / (512 / column tiles)) all workgroups are resident at once and finish together. rows_per_wg = BM is the plain one-tile form
// (windowed convs, whose main loop is long: a partial tile would cost a whole one).
// fp8 operand images (fp8_ops.hip): [rows][64 B] with 16-byte chunk c of row r in slot c ^ ((r>>2)&3) - DmaKC's image at one byte per element. For
// v_mfma_scale_f32_32x32x64_f8f6f4 lane (r, h) takes the chunks h and 2 + h of row r (intrin.h): one K tile = ONE instruction per 32 x 32 block.
DEV int fp8_frag_off64(int x0, int half, int lane) {
  const int r = x0 + (lane & 31);
  return r * 64 + (((2 * half + (lane >> 5)) ^ ((r >> 2) & 3)) << 4);
}

// F8 (clite_conv_dgrad_fp8): T stays bf16 (the epilogue's tensors), the operands are one-byte images - A = the gradient in e5m2, B = the transposed
// weights in e4m3 - multiplied by the block-scaled instruction at unit scales; the accumulators are de-quantised (f8_a[1] * f8_b[1], the inverse
// per-tensor scales) before the epilogue, which is the bf16 kernel's.
// AFF (probe builds only, -DCLITE_PROBE_AFRAG; tools/probe_afrag.py): the A operand is a raw convolution output and relu(a * scale[k] + shift[k]) — the
// BatchNorm in front of a 1 x 1 convolution — is applied to each A fragment between the LDS read and the MFMA; f8_a / f8_b carry scale / shift (f32 [K]),
// staged once per workgroup behind the operand ring. A lane's fragment holds 8 consecutive k of one row, so each k-step reads 8 + 8 coefficients (two
// half-wave broadcast reads each).
template <typename T, class CFG, class LA, class LB, int FORM = 0, bool SPLIT = false, bool F8 = false, bool AFF = false>
__global__ __launch_bounds__(256, (FORM && CLITE_BN_HALF) ? 3 : 2) void igemm_dma_bn_kernel(LA la, LB lb, Epilogue ep, RowMap rm, int M, int N, int ktiles, int rows_per_wg,
                                                                                            const float* f8_a, const float* f8_b) {
  constexpr int BM = CFG::BM, BN = CFG::BN, BK = CFG::BK;
  constexpr int RM = CFG::RM, RN = CFG::RN;
#ifndef CLITE_BN_STAGES
#define CLITE_BN_STAGES 3
#endif
  constexpr int NSTAGE = CLITE_BN_STAGES;
  constexpr int STAGE = LA::BYTES + LB::BYTES;
  constexpr int EPIB = (CFG::BM / ((FORM && CLITE_BN_HALF) ? 2 : 1)) * CFG::EPI_PITCH;      // the epilogue stages the whole tile, or half of it at a time (igemm_epilogue_bn NP)
  constexpr int RED = (256 / (CFG::BN / 8)) * (CFG::BN / 8) * 16 * 4;      // bn_epi_finish's fold image
  constexpr int SMEM0 = (NSTAGE * STAGE > EPIB) ? NSTAGE * STAGE : EPIB;
  constexpr int SMEM1 = SMEM0 > RED ? SMEM0 : RED;
  constexpr int AFFB = AFF ? 2 * 512 * 4 : 0;                 // scale | shift of up to 512 input channels
  constexpr int SMEM = SMEM1 + AFFB;
  constexpr int LOADS_PER_TILE = LA::NI + LB::NI;            // per wave
  static_assert(sizeof(T) == 2 || sizeof(T) == 4, "bf16 or f32");
  static_assert(!AFF || (sizeof(T) == 2 && !F8), "fragment-side affine: bf16 operands");
  // (Measured and rejected, round 3: touching every 128-byte line of the tile's BatchNorm-input / residual rows by LDS-DMA into a scratch at
  // the start of the tile, so that the epilogue's loads hit L2 — same-box A/B 16.99 -> 17.20 ms: the extra requests compete with the
  // operand ring for the same L2 -> CU path that bounds the main loop.)
  __shared__ __attribute__((aligned(1024))) char smem[SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = wave_uniform(tid >> 6);
  const int wm0 = (wave / CFG::WAVES_N) * CFG::WM;
  const int wn0 = (wave % CFG::WAVES_N) * CFG::WN;
  const int tiles_n = (N + BN - 1) / BN;
  // XCD-aware order (igemm_dma_kernel): each XCD gets a contiguous run of (row range, column tile) pairs, column tiles of one row range adjacent
  const int nwg = gridDim.x, xcd = blockIdx.x & 7, xq = nwg >> 3, xr = nwg & 7;
  const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
  const int slice = wg / tiles_n;
  const int n0 = (wg - slice * tiles_n) * BN;
  const int row_begin = slice * rows_per_wg;
  int row_end = row_begin + rows_per_wg;
  if (row_end > M) row_end = M;
  la.g.rows = row_end;                      // rows past this workgroup's range gather as out of range (zeros)

  int aoff[RM][BK / 16 > 0 ? BK / 16 : 1], boff[RN][BK / 16 > 0 ? BK / 16 : 1];
  if constexpr (F8) {
    static_assert(BK == 64 && sizeof(T) == 2, "fp8 operands: 64-byte K rows, bf16 epilogue");
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int i = 0; i < RM; ++i) aoff[i][q] = fp8_frag_off64(wm0 + i * 32, q, lane);
#pragma unroll
      for (int j = 0; j < RN; ++j) boff[j][q] = fp8_frag_off64(wn0 + j * 32, q, lane);
    }
  } else if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
#pragma unroll
      for (int i = 0; i < RM; ++i) aoff[i][ks] = LA::frag_off(wm0 + i * 32, ks, lane);
#pragma unroll
      for (int j = 0; j < RN; ++j) boff[j][ks] = LB::frag_off(wn0 + j * 32, ks, lane);
    }
  }

#if CLITE_STAMP
  if (tid == 0) { for (int i = 0; i < 7; ++i) clite_ph[i] = 0; clite_ph[7] = __builtin_amdgcn_s_memrealtime(); }
#endif
  if constexpr (AFF) {
    float* tbl = (float*)(smem + SMEM1);
    for (int k = tid; k < ktiles * BK; k += 256) { tbl[k] = f8_a[k]; tbl[512 + k] = f8_b[k]; }
    // (the first tile's barrier_raw orders these stores before the first fragment's reads)
  }
  BnEpiState est;
  bn_epi_begin<CFG>(est, ep, N, n0, tid);
  PHASE(0);          // launch prologue: addresses, BatchNorm means
  // rows of epilogue operands requested ahead of the tile's main loop (BnRows): the specialised forms with whole-tile staging
  constexpr int EARLY = (FORM && !CLITE_BN_HALF && sizeof(T) == 2) ? (CLITE_BN_EARLY < BnRows<T, CFG, FORM>::ROWS_PT ? CLITE_BN_EARLY : BnRows<T, CFG, FORM>::ROWS_PT) : 0;
  for (int m0 = row_begin; m0 < row_end; m0 += BM) {
    BnRows<T, CFG, FORM> rows;
#pragma unroll
    for (int q = 0; q < EARLY; ++q) rows.request(q, ep, rm, row_end, N, m0, n0, tid);
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
#pragma unroll
    for (int pz = 0; pz < NSTAGE - 1; ++pz) {
#if CLITE_ABLATE != 2
      if (pz < ktiles) {
        DmaIssue<LA>::go(la, sa, smem + pz * STAGE, wave, lane, m0);
        DmaIssue<LB>::go(lb, sb, smem + pz * STAGE + LA::BYTES, wave, lane, n0);
      }
#endif
    }
    int buf = 0;
    PHASE(1);        // per tile: early requests, loader state, ring prefill issued
    for (int t = 0; t < ktiles; ++t) {
      const int after = ktiles - 1 - t;      // tile t has landed once at most min(NSTAGE - 2, tiles after t) younger tiles are outstanding
      if (NSTAGE >= 5 && after >= 3) wait_vmcnt<3 * LOADS_PER_TILE>();
      else if (NSTAGE >= 4 && after >= 2) wait_vmcnt<2 * LOADS_PER_TILE>();
      else if (after >= 1) wait_vmcnt<LOADS_PER_TILE>();
      else wait_vmcnt<0>();
      barrier_raw();
      if (t == 0) PHASE(2);      // wait for the first operand tile
      const char* abuf = smem + buf * STAGE;
      const char* bbuf = abuf + LA::BYTES;
      if constexpr (F8) {
        u32x4 alo[RM], ahi[RM], blo[RN], bhi[RN];
#pragma unroll
        for (int i = 0; i < RM; ++i) { alo[i] = *(const u32x4*)(abuf + aoff[i][0]); ahi[i] = *(const u32x4*)(abuf + aoff[i][1]); }
#pragma unroll
        for (int j = 0; j < RN; ++j) { blo[j] = *(const u32x4*)(bbuf + boff[j][0]); bhi[j] = *(const u32x4*)(bbuf + boff[j][1]); }
        if (t + NSTAGE - 1 < ktiles) {
          int nb = buf + NSTAGE - 1; if (nb >= NSTAGE) nb -= NSTAGE;
          DmaIssue<LA>::go(la, sa, smem + nb * STAGE, wave, lane, m0);
          DmaIssue<LB>::go(lb, sb, smem + nb * STAGE + LA::BYTES, wave, lane, n0);
        }
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
          for (int j = 0; j < RN; ++j) acc[i][j] = mfma32x64_bf8_fp8(alo[i], ahi[i], blo[j], bhi[j], acc[i][j]);
      } else if constexpr (sizeof(T) == 2) {
        bf16x8 af0[RM], bf0[RN];
#pragma unroll
        for (int i = 0; i < RM; ++i) af0[i] = LA::frag_at(abuf + aoff[i][0]);
#pragma unroll
        for (int j = 0; j < RN; ++j) bf0[j] = LB::frag_at(bbuf + boff[j][0]);
#if CLITE_ABLATE != 2
        if (t + NSTAGE - 1 < ktiles) {
          int nb = buf + NSTAGE - 1; if (nb >= NSTAGE) nb -= NSTAGE;
          DmaIssue<LA>::go(la, sa, smem + nb * STAGE, wave, lane, m0);
          DmaIssue<LB>::go(lb, sb, smem + nb * STAGE + LA::BYTES, wave, lane, n0);
        }
#endif
#pragma unroll
        for (int ks = 0; ks < (CLITE_ABLATE == 1 ? 0 : BK / 16); ++ks) {
          bf16x8 af[RM], bfr[RN];
#pragma unroll
          for (int i = 0; i < RM; ++i) af[i] = ks == 0 ? af0[i] : LA::frag_at(abuf + aoff[i][ks]);
#pragma unroll
          for (int j = 0; j < RN; ++j) bfr[j] = ks == 0 ? bf0[j] : LB::frag_at(bbuf + boff[j][ks]);
          if constexpr (AFF) {
            const float* sc = (const float*)(smem + SMEM1) + t * BK + (ks * 2 + (lane >> 5)) * 8;
            const f32x4 s0 = *(const f32x4*)sc, s1 = *(const f32x4*)(sc + 4), h0 = *(const f32x4*)(sc + 512), h1 = *(const f32x4*)(sc + 516);
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float v = fmaf(bf2f(af[i][e]), e < 4 ? s0[e & 3] : s1[e & 3], e < 4 ? h0[e & 3] : h1[e & 3]);
                af[i][e] = f2bf(fmaxf(v, 0.f));
              }
          }
#pragma unroll
          for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j) acc[i][j] = mfma32_bf16(af[i], bfr[j], acc[i][j]);
        }
      } else {
        if (t + NSTAGE - 1 < ktiles) {
          int nb = buf + NSTAGE - 1; if (nb >= NSTAGE) nb -= NSTAGE;
          DmaIssue<LA>::go(la, sa, smem + nb * STAGE, wave, lane, m0);
          DmaIssue<LB>::go(lb, sb, smem + nb * STAGE + LA::BYTES, wave, lane, n0);
        }
        if constexpr (SPLIT) {
          mma_tile_f32_split<CFG, LA, LB>(abuf, bbuf, wm0, wn0, lane, acc);
        } else {
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
      }
      if (++buf == NSTAGE) buf = 0;
    }
    barrier_raw();          // every wave is past its last fragment read before the epilogue reuses the LDS
    PHASE(3);        // main loop
    if constexpr (F8) {
      const float dq = f8_a[1] * f8_b[1];
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] *= dq;
    }
    igemm_epilogue_bn<T, CFG, FORM, EARLY>(acc, est, rows, ep, rm, smem, row_end, N, m0, n0, tid, lane, wave, wm0, wn0);
    // (igemm_epilogue_bn ends with a barrier: every wave is past its last read of the epilogue's LDS image before the next tile's operands land)
  }
  bn_epi_finish<CFG>(est, ep, smem, N, n0, tid);
#if CLITE_STAMP
  PHASE(6);          // column-sum fold + atomics
  if (tid == 0 && blockIdx.x < 65536) { for (int i = 0; i < 7; ++i) clite_dbg[blockIdx.x * 8 + i] = clite_ph[i]; clite_dbg[blockIdx.x * 8 + 7] = (unsigned long long)((row_end - row_begin + BM - 1) / BM); }
#endif
}


// ---- two K-groups per workgroup -------------------------------------------------------------------------------------------
// Small grids (at most one workgroup per CU: the BERT GEMMs with N = 768, the 7x7-resolution convs) leave each SIMD with a single
// wave whose DMA issue, LDS reads and MFMAs serialise. Here a workgroup has 8 waves = 2 groups of 4; group g runs the pipeline
// above over its own half of the K tiles with its own LDS ring (an in-workgroup split-K that shares only the barriers), so every
// SIMD hosts two independent MFMA streams. Group 1 then hands its accumulators to group 0 through LDS and group 0 runs the fused
// epilogue; group 1 only keeps the barrier count.
template <typename T, class CFG, class LA, class LB, int NSTAGE>
__global__ __launch_bounds__(512) void igemm_dma_kernel_g2(LA la, LB lb, Epilogue ep, RowMap rm, int M, int N, int ktiles, int ktiles_per_split, int xsplits) {
  constexpr int BM = CFG::BM, BN = CFG::BN, BK = CFG::BK;
  constexpr int RM = CFG::RM, RN = CFG::RN;
  constexpr int STAGE = LA::BYTES + LB::BYTES;
  constexpr int RING = NSTAGE * STAGE;
  constexpr int XCHG = BM * BN * 4;                         // group 1's accumulators, f32, thread-major
  constexpr int NEED = 2 * RING > XCHG ? 2 * RING : XCHG;
  constexpr int SMEM = NEED > CFG::EPI_BYTES ? NEED : CFG::EPI_BYTES;
  constexpr int LOADS_PER_TILE = LA::NI + LB::NI;
  __shared__ __attribute__((aligned(1024))) char smem[SMEM];

  const int tid = threadIdx.x & 255;
  const int lane = tid & 63;
  const int wave = wave_uniform(tid >> 6);                  // wave index inside the group
  const int group = wave_uniform(threadIdx.x >> 8);
  const int wm0 = (wave / CFG::WAVES_N) * CFG::WM;
  const int wn0 = (wave % CFG::WAVES_N) * CFG::WN;

  const int tiles_n = (N + BN - 1) / BN;
  int wg, zsplit;
  if (xsplits > 0) {
    // split-K with >= 8 splits (weight gradients): all output tiles of one K range run on ONE XCD (workgroups b and b+8 share an XCD),
    // so each operand panel is fetched into one L2 instead of eight; XCD x takes the splits x, x+8, ...
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
  const int s_begin = zsplit * ktiles_per_split;
  int s_end = s_begin + ktiles_per_split;
  if (s_end > ktiles) s_end = ktiles;
  const int half = (s_end - s_begin + 1) >> 1;              // group 0 takes the first `half` tiles, group 1 the rest
  const int t_begin = s_begin + group * half;
  const int t_end = group == 0 ? s_begin + half : s_end;
  char* ring = smem + group * RING;

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

  int aoff[RM][BK / 16 > 0 ? BK / 16 : 1], boff[RN][BK / 16 > 0 ? BK / 16 : 1];
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
#pragma unroll
      for (int i = 0; i < RM; ++i) aoff[i][ks] = LA::frag_off(wm0 + i * 32, ks, lane);
#pragma unroll
      for (int j = 0; j < RN; ++j) boff[j][ks] = LB::frag_off(wn0 + j * 32, ks, lane);
    }
  }

#pragma unroll
  for (int pz = 0; pz < NSTAGE - 1; ++pz) {
    if (t_begin + pz < t_end) {
      DmaIssue<LA>::go(la, sa, ring + pz * STAGE, wave, lane, m0);
      DmaIssue<LB>::go(lb, sb, ring + pz * STAGE + LA::BYTES, wave, lane, n0);
    }
  }
  int buf = 0;
  for (int it = 0; it < half; ++it) {                       // both groups run `half` iterations: the barriers are workgroup-wide
    const int t = t_begin + it;
    const bool live = t < t_end;
    const int after = t_end - 1 - t;
    if (NSTAGE >= 5 && after >= 3) wait_vmcnt<3 * LOADS_PER_TILE>();
    else if (NSTAGE >= 4 && after >= 2) wait_vmcnt<2 * LOADS_PER_TILE>();
    else if (after >= 1) wait_vmcnt<LOADS_PER_TILE>();
    else wait_vmcnt<0>();
    barrier_raw();
    const char* abuf = ring + buf * STAGE;
    const char* bbuf = abuf + LA::BYTES;
    bf16x8 af0[RM], bf0[RN];
    if constexpr (sizeof(T) == 2) {        // first k-step's fragments before the DMA issue (see igemm_dma_kernel)
      if (live) {
#pragma unroll
        for (int i = 0; i < RM; ++i) af0[i] = LA::frag_at(abuf + aoff[i][0]);
#pragma unroll
        for (int j = 0; j < RN; ++j) bf0[j] = LB::frag_at(bbuf + boff[j][0]);
      }
    }
    if (t + NSTAGE - 1 < t_end) {
      int nb = buf + NSTAGE - 1; if (nb >= NSTAGE) nb -= NSTAGE;
      DmaIssue<LA>::go(la, sa, ring + nb * STAGE, wave, lane, m0);
      DmaIssue<LB>::go(lb, sb, ring + nb * STAGE + LA::BYTES, wave, lane, n0);
    }
    if (live) {
      if constexpr (sizeof(T) == 2) {
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
      } else {
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
    }
    if (++buf == NSTAGE) buf = 0;
  }
  lds_barrier();          // (no DMA is outstanding any more) every wave is past its last fragment read: the LDS is free
  // group 1 -> group 0: element (i, j, r) of thread tid at ((i*RN + j)*16 + r)*256 + tid  (consecutive lanes, consecutive words)
  float* xchg = (float*)smem;
  if (group == 1) {
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
      for (int j = 0; j < RN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) xchg[((i * RN + j) * 16 + r) * 256 + tid] = acc[i][j][r];
  }
  lds_barrier();
  if (group == 0) {
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
      for (int j = 0; j < RN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] += xchg[((i * RN + j) * 16 + r) * 256 + tid];
  }
  lds_barrier();
  if (group == 0) {
    igemm_epilogue<T, CFG>(acc, ep, rm, smem, M, N, m0, n0, tid, lane, wave, wm0, wn0);
  } else if (!ep.atomic) {   // keep the workgroup barrier count of igemm_epilogue
    for (int pass = 0; pass < CFG::WAVES_M; ++pass) { lds_barrier(); lds_barrier(); }
    if (ep.colsum) lds_barrier();
  }
}

}  // namespace clite
#endif  // CLITE_IGEMM_DMA_H
