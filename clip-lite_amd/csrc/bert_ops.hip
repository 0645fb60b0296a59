// Kernels of the BERT text encoder outside the GEMMs (reference encoder.py:165-196 -> transformers.BertModel,
// BertConfig defaults: hidden 768, 12 heads x 64, LayerNorm eps 1e-12, dropout 0.1, additive attention mask,
// pooler = tanh(W h_CLS + b)): embedding gather, LayerNorm forward/backward (also nn.LayerNorm of the MI heads,
// loss.py:23), per-head attention forward/backward for captions of <= 32 tokens (config.py:69: 30).
#include "vec.h"
#include "rng.h"
#include "det.h"
#include "clite.h"

using namespace clite;

namespace {

constexpr int LN_MAXCH = 4;   // 8-element chunks per lane: C <= 64*4*8 = 2048

struct Drop { float p; uint64_t seed; uint32_t site; };

DEV void apply_dropout8(const Drop& d, size_t idx, float (&v)[8]) {
  float u[8];
  dropout_uniform8(d.seed, d.site, idx, u);
  float ks = 1.0f / (1.0f - d.p);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = u[e] >= d.p ? v[e] * ks : 0.f;
}

// one wave per row; out = dropout((x - mean) * rstd * gamma + beta); stats[row] = (mean, rstd)
// Q (bf16, clite_layernorm_fwd_q8): the producer-fused e4m3 quantiser of the fp8 linears that read `out` - the e4m3 copy of the stored value at a
// delayed scale and the call's max |out| (one integer atomic max per workgroup), as bn_apply's (resnet_ops.hip)
struct LnQ8 { uint8_t* out; const float* scale; float* amax; };
template <typename T, int NCH, bool Q = false>      // NCH = 8-element chunks per lane: C <= 64*NCH*8 (2 covers BERT's 768, 4 the 2048-wide heads)
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* x, const float* gamma, const float* beta, float eps, T* out,
                                                            float* stats, int M, int C, Drop drop, LnQ8 q8) {
  float qmax = 0.f;
  bool qnan = false;
  const float qscale = (Q && q8.out) ? q8.scale[0] : 1.f;
  seed_resolve(drop.seed, drop.site);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = C / 8;
  const float invC = 1.0f / (float)C;
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int c = lane + 64 * i;
      if (c < nchunk) {
        load8(x + (size_t)row * C + c * 8, v[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[i][e];
      }
    }
    float mean = wave_sum(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int c = lane + 64 * i;
      if (c < nchunk) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { float d = v[i][e] - mean; q += d * d; }
      }
    }
    float rstd = rsqrtf(wave_sum(q) * invC + eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int c = lane + 64 * i;
      if (c < nchunk) {
        float g[8], b[8], o[8];
        load8(gamma + c * 8, g);
        load8(beta + c * 8, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
        size_t idx = (size_t)row * C + c * 8;
        if (drop.p > 0.f) apply_dropout8(drop, idx, o);
        store8(out + idx, o);
        if constexpr (Q) {
          round8_bf16(o);          // quantise / measure the value as stored
#pragma unroll
          for (int e = 0; e < 8; ++e) { qmax = fmaxf(qmax, fabsf(o[e])); qnan = qnan || o[e] != o[e]; }
          if (q8.out) {
            uint32_t w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float p0 = o[2 * e] * qscale, q0 = o[2 * e + 1] * qscale;          // (a NaN passes the clamp untouched: fp8_ops.hip)
              const float pp = p0 != p0 ? p0 : fminf(fmaxf(p0, -448.f), 448.f), qq = q0 != q0 ? q0 : fminf(fmaxf(q0, -448.f), 448.f);
              w[e] = cvt2_fp8(pp, qq);
            }
            *(u32x2*)(q8.out + idx) = u32x2{w[0] | (w[1] << 16), w[2] | (w[3] << 16)};
          }
        }
      }
    }
    if (lane == 0 && stats) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
  }
  if constexpr (Q) {
    if (q8.amax) {
      __shared__ uint32_t red[4];
      uint32_t mb = qnan ? 0x7FC00000u : f32_bits(qmax);
#pragma unroll
      for (int sh = 32; sh >= 1; sh >>= 1) { const uint32_t o = (uint32_t)wave_shfl_xor_i((int)mb, sh); mb = o > mb ? o : mb; }
      if (lane == 0) red[wave] = mb;
      __syncthreads();
      if (threadIdx.x == 0) {
        uint32_t b = red[0];
        for (int w = 1; w < 4; ++w) b = red[w] > b ? red[w] : b;
        atomic_max_u32((uint32_t*)q8.amax + (blockIdx.x % CLITE_FP8_AMAX_REPLICAS) * CLITE_FP8_AMAX_STRIDE, b);
      }
    }
  }
}

// dy' = dropout_mask_in(dy); dx = rstd*(g - mean(g) - xhat*mean(g*xhat)), g = dy'*gamma; dgamma += sum dy'*xhat; dbeta += sum dy'
// optional second output dx_masked = dropout_mask_out(dx)
// NW waves per workgroup, one row per wave at a time; one float atomic per column per workgroup for dgamma / dbeta. What the kernel costs at
// M = 3840, C = 768 (tools/probe_ln.py): 11 us without the parameter gradients, 13 with them, 20 with the recomputed input dropout mask and
// 27 with the masked second output as well — the Philox calls, not the atomics, are the expensive part, so the grid keeps one row or two
// per wave (256 workgroups of 8 waves) rather than trading waves for fewer adders.
template <typename T, int NCH, int NW>
__global__ __launch_bounds__(NW * 64) void layernorm_bwd_kernel(const T* dy, const T* x, const float* stats, const float* gamma, T* dx, T* dx_masked,
                                                                float* dgamma, float* dbeta, float* dcolsum, int M, int C, Drop drop_in, Drop drop_out) {
  seed_resolve(drop_in.seed, drop_in.site);
  seed_resolve(drop_out.seed, drop_out.site);
  __shared__ float red[NW][64 * NCH * 8];   // [wave][per-lane partials], used once for dgamma and once for dbeta
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = C / 8;
  const float invC = 1.0f / (float)C;
  float ag[NCH][8], ab[NCH][8], ax[NCH][8];      // ax: column sums of the stored gradient = bias gradient of the Linear that fed this LayerNorm
#pragma unroll
  for (int i = 0; i < NCH; ++i) { zero8(ag[i]); zero8(ab[i]); zero8(ax[i]); }
  for (int row = blockIdx.x * NW + wave; row < M; row += gridDim.x * NW) {
    float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    float d[NCH][8], xh[NCH][8];          // g = d*gamma is re-formed in the second loop (gamma is L1-resident): 8*NCH fewer live registers
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int c = lane + 64 * i;
      if (c < nchunk) {
        size_t idx = (size_t)row * C + c * 8;
        load8(dy + idx, d[i]);
        if (drop_in.p > 0.f) apply_dropout8(drop_in, idx, d[i]);
        float xv[8], gm[8];
        load8(x + idx, xv);
        load8(gamma + c * 8, gm);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xh[i][e] = (xv[e] - mean) * rstd;
          const float ge = d[i][e] * gm[e];
          s1 += ge;
          s2 += ge * xh[i][e];
          ag[i][e] += d[i][e] * xh[i][e];
          ab[i][e] += d[i][e];
        }
      }
    }
    float m1 = wave_sum(s1) * invC, m2 = wave_sum(s2) * invC;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int c = lane + 64 * i;
      if (c < nchunk) {
        size_t idx = (size_t)row * C + c * 8;
        float o[8], gm[8];
        load8(gamma + c * 8, gm);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rstd * (d[i][e] * gm[e] - m1 - xh[i][e] * m2);
        store8(dx + idx, o);
        if (dx_masked) {
          if (drop_out.p > 0.f) apply_dropout8(drop_out, idx, o);
          store8(dx_masked + idx, o);
        }
        if (dcolsum) {      // sums of what the consumer reads: the masked tensor when there is one, rounded to the storage type
          if (sizeof(T) == 2) round8_bf16(o);
#pragma unroll
          for (int e = 0; e < 8; ++e) ax[i][e] += o[e];
        }
      }
    }
  }
  // fold the 4 waves' partial dgamma/dbeta through LDS (two passes to bound LDS), one atomic per column per block
  for (int which = 0; which < 3; ++which) {
    if (which == 2 && !dcolsum) break;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[wave][(i * 64 + lane) * 8 + e] = which == 0 ? ag[i][e] : which == 1 ? ab[i][e] : ax[i][e];
    __syncthreads();
    float* dst = which == 0 ? dgamma : which == 1 ? dbeta : dcolsum;
    if (dst) {
      for (int j = threadIdx.x; j < NCH * 64 * 8; j += NW * 64) {
        int i = j / 512, l = (j / 8) % 64, e = j % 8;
        int c = l + 64 * i;
        float sacc = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sacc += red[w][j];
        if (c < nchunk) atomic_add_f32(dst + c * 8 + e, sacc);
      }
    }
  }
}

// s0[row] = word[ids[row]] + pos[row % L] + type[0]
template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* ids, const T* word, const T* pos, const T* type, T* out, int M, int L, int C, int vocab) {
  const int nchunk = C / 8;
  size_t total = (size_t)M * nchunk;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    int c = (int)(i % nchunk);
    int row = (int)(i / nchunk);
    int64_t id = ids[row];
    if (id < 0) id = 0;
    if (id >= vocab) id = vocab - 1;
    float a[8], b[8], t[8];
    load8(word + (size_t)id * C + c * 8, a);
    load8(pos + (size_t)(row % L) * C + c * 8, b);
    load8(type + c * 8, t);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += b[e] + t[e];
    store8(out + (size_t)row * C + c * 8, a);
  }
}

// dword[ids[row]] += d[row] for ids[row] != padding_idx (nn.Embedding(padding_idx) semantics), dpos[l] += sum_b d[b*L + l].
// Workgroup = (position l, batch segment): a thread owns one 8-column chunk and walks the segment's captions at that position, so
// (a) the position sum is a register accumulation, and (b) consecutive rows with the SAME token id — [CLS] at position 0 and [SEP] at the
// last position of every caption — are summed in registers and leave as one atomic set per run instead of one per caption. (A kernel
// with one atomic per row spent 100 us here: thousands of adders on the [CLS] / [SEP] rows serialise at the memory side.)
constexpr int EMB_SEGS = 4, EMB_COLS = 16;      // C <= 128 * EMB_COLS
DEV float ld1(const bf16* p) { return bf2f(*p); }
DEV float ld1(const float* p) { return *p; }
template <typename T>
__global__ __launch_bounds__(128) void embed_bwd_kernel(const int64_t* __restrict__ ids, const T* __restrict__ d, float* __restrict__ dword, float* __restrict__ dpos,
                                                        int B, int L, int C, int vocab, int padding_idx) {
  // (position, segment) pairs are walked grid-stride: normally one per workgroup; the deterministic-reduction mode (det.h) launches a single
  // workgroup, whose threads own disjoint columns and add in program order
  for (int blk = blockIdx.x; blk < L * EMB_SEGS; blk += gridDim.x) {
  const int l = blk / EMB_SEGS, seg = blk % EMB_SEGS;
  const int per = (B + EMB_SEGS - 1) / EMB_SEGS;
  const int b0 = seg * per, b1 = b0 + per < B ? b0 + per : B;
  const int t = threadIdx.x;
  // lane t owns columns t, t + 128, ...: a wave's float atomics then cover 256 contiguous bytes (the full-rate shape), and its 2-byte loads
  // 128 contiguous bytes (the rows are small: 6 MB in all)
  float pacc[EMB_COLS], wacc[EMB_COLS];
#pragma unroll
  for (int j = 0; j < EMB_COLS; ++j) { pacc[j] = 0.f; wacc[j] = 0.f; }
  int64_t cur = -1;
  auto flush = [&]() {
    if (dword && cur >= 0 && cur != padding_idx) {
#pragma unroll
      for (int j = 0; j < EMB_COLS; ++j)
        if (t + 128 * j < C) atomic_add_f32(dword + (size_t)cur * C + t + 128 * j, wacc[j]);
    }
#pragma unroll
    for (int j = 0; j < EMB_COLS; ++j) wacc[j] = 0.f;
  };
  // the next caption's row is fetched before this one's atomics are issued (the loop is otherwise one load latency per caption)
  float v[EMB_COLS], vn[EMB_COLS];
  int64_t id = 0, idn = 0;
  auto fetch = [&](int b, float (&dst)[EMB_COLS], int64_t& di) {
    const size_t row = (size_t)b * L + l;
    di = ids[row];
#pragma unroll
    for (int j = 0; j < EMB_COLS; ++j) dst[j] = t + 128 * j < C ? ld1(d + row * C + t + 128 * j) : 0.f;
  };
  if (b0 < b1) fetch(b0, vn, idn);
  for (int b = b0; b < b1; ++b) {
    id = idn;
#pragma unroll
    for (int j = 0; j < EMB_COLS; ++j) v[j] = vn[j];
    if (b + 1 < b1) fetch(b + 1, vn, idn);
    if (id < 0) id = 0;
    if (id >= vocab) id = vocab - 1;
    if (id != cur) { flush(); cur = id; }
#pragma unroll
    for (int j = 0; j < EMB_COLS; ++j) { pacc[j] += v[j]; wacc[j] += v[j]; }
  }
  flush();
  if (dpos && b1 > b0) {
#pragma unroll
    for (int j = 0; j < EMB_COLS; ++j)
      if (t + 128 * j < C) atomic_add_f32(dpos + (size_t)l * C + t + 128 * j, pacc[j]);
  }
  }
}

// ------------------------------------------------------------------------------------------------ attention, L <= 32
// One wave per (batch, head). qkv: [B*L][3*H*64] (q | k | v, head h at columns h*64..). ctx: [B*L][H*64].
constexpr int AT_L = 32, AT_D = 64, AT_PQ = AT_D + 1, AT_PP = AT_L + 1;
#define AT_NEG (-3.4028234663852886e38f)

struct AttnSmem {
  float q[AT_L * AT_PQ], k[AT_L * AT_PQ], v[AT_L * AT_PQ], p[AT_L * AT_PP];
  float m[AT_L * AT_PP];   // dropout multiplier of p[i][j]: 0 or 1/(1-p)
};

// m[i][j] for one (batch, head): element index ((bh*32 + i)*32 + j); each lane draws 4 consecutive j per Philox call
DEV void attn_dropmask(AttnSmem& sm, const Drop& drop, int bh, int lane) {
  const float ks = drop.p > 0.f ? 1.0f / (1.0f - drop.p) : 1.f;
  for (int q4 = lane; q4 < AT_L * AT_L / 4; q4 += 64) {
    int i = q4 >> 3, j0 = (q4 & 7) * 4;
    float u[4] = {1.f, 1.f, 1.f, 1.f};
    if (drop.p > 0.f) rng_uniform4(drop.seed, drop.site, ((size_t)bh * AT_L + i) * AT_L + j0, u);
#pragma unroll
    for (int e = 0; e < 4; ++e) sm.m[i * AT_PP + j0 + e] = u[e] >= drop.p ? ks : 0.f;
  }
}

template <typename T>
DEV void attn_load_rows(const T* base, size_t row_stride, int L, float* dst, int lane) {
  // L rows x 64 values; 8 lanes per row, 8 rows per pass
  for (int r0 = 0; r0 < AT_L; r0 += 8) {
    int r = r0 + (lane >> 3), c = (lane & 7) * 8;
    float v[8];
    if (r < L) load8(base + (size_t)r * row_stride + c, v); else zero8(v);
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[r * AT_PQ + c + e] = v[e];
  }
}

// scores + softmax for all rows: p[i][j] (un-dropped probabilities), i,j < 32 (rows/cols >= L are zero)
DEV void attn_probs(AttnSmem& sm, const int64_t* mask_row, int L, int lane) {
  const int j = lane & 31, half = lane >> 5;
  const bool jvalid = j < L;
  float madd = 0.f;
  if (jvalid && mask_row) madd = mask_row[j] != 0 ? 0.f : AT_NEG;
  for (int i0 = 0; i0 < AT_L; i0 += 2) {
    int i = i0 + half;
    float s = 0.f;
#pragma unroll 16
    for (int d = 0; d < AT_D; ++d) s += sm.q[i * AT_PQ + d] * sm.k[j * AT_PQ + d];
    s = s * 0.125f + madd;
    if (!jvalid) s = -INFINITY;
    float mx = s;
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) mx = fmaxf(mx, wave_shfl_xor(mx, m));
    float e = jvalid ? __expf(s - mx) : 0.f;
    float den = e;
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) den += wave_shfl_xor(den, m);
    sm.p[i * AT_PP + j] = (i < L) ? e / den : 0.f;
  }
}

template <typename T>
__global__ __launch_bounds__(64) void attention_fwd_kernel(const T* qkv, const int64_t* mask, T* ctx, int B, int L, int H, Drop drop) {
  seed_resolve(drop.seed, drop.site);
  __shared__ AttnSmem sm;
  const int lane = threadIdx.x;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const size_t ld = (size_t)3 * H * AT_D;
  const T* base = qkv + (size_t)b * L * ld + h * AT_D;
  attn_load_rows(base, ld, L, sm.q, lane);
  attn_load_rows(base + H * AT_D, ld, L, sm.k, lane);
  attn_load_rows(base + 2 * H * AT_D, ld, L, sm.v, lane);
  __syncthreads();
  attn_probs(sm, mask ? mask + (size_t)b * L : nullptr, L, lane);
  attn_dropmask(sm, drop, b * H + h, lane);
  __syncthreads();
  for (int i = 0; i < L; ++i) {
    float o = 0.f;
    for (int j = 0; j < L; ++j) o += sm.p[i * AT_PP + j] * sm.m[i * AT_PP + j] * sm.v[j * AT_PQ + lane];
    size_t oi = ((size_t)b * L + i) * ((size_t)H * AT_D) + h * AT_D + lane;
    if constexpr (sizeof(T) == 2) ctx[oi] = f2bf(o); else ctx[oi] = o;
  }
}

struct AttnBwdSmem {
  AttnSmem f;
  float dO[AT_L * AT_PQ], dS[AT_L * AT_PP];
};

template <typename T>
__global__ __launch_bounds__(64) void attention_bwd_kernel(const T* qkv, const int64_t* mask, const T* dctx, T* dqkv, int B, int L, int H, Drop drop) {
  seed_resolve(drop.seed, drop.site);
  __shared__ AttnBwdSmem sm;
  const int lane = threadIdx.x;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const size_t ld = (size_t)3 * H * AT_D;
  const T* base = qkv + (size_t)b * L * ld + h * AT_D;
  attn_load_rows(base, ld, L, sm.f.q, lane);
  attn_load_rows(base + H * AT_D, ld, L, sm.f.k, lane);
  attn_load_rows(base + 2 * H * AT_D, ld, L, sm.f.v, lane);
  attn_load_rows(dctx + (size_t)b * L * ((size_t)H * AT_D) + h * AT_D, (size_t)H * AT_D, L, sm.dO, lane);
  __syncthreads();
  attn_probs(sm.f, mask ? mask + (size_t)b * L : nullptr, L, lane);
  attn_dropmask(sm.f, drop, b * H + h, lane);
  __syncthreads();
  const int j = lane & 31, half = lane >> 5;
  // dP[i][j] = keep[i][j]*ks * sum_d dO[i][d] V[j][d];  dS = P * (dP - sum_j dP*P)
  for (int i0 = 0; i0 < AT_L; i0 += 2) {
    int i = i0 + half;
    float dp = 0.f;
#pragma unroll 16
    for (int d = 0; d < AT_D; ++d) dp += sm.dO[i * AT_PQ + d] * sm.f.v[j * AT_PQ + d];
    dp *= sm.f.m[i * AT_PP + j];
    float pij = sm.f.p[i * AT_PP + j];
    float t = dp * pij;
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) t += wave_shfl_xor(t, m);
    sm.dS[i * AT_PP + j] = pij * (dp - t);
  }
  __syncthreads();
  // lane = d.  dV[j][d] = sum_i Pd[i][j] dO[i][d];  dK[j][d] = sum_i dS[i][j] Q[i][d] / 8;  dQ[i][d] = sum_j dS[i][j] K[j][d] / 8
  T* obase = dqkv + (size_t)b * L * ld + h * AT_D;
  for (int r = 0; r < L; ++r) {
    float dq = 0.f, dk = 0.f, dv = 0.f;
    for (int t = 0; t < L; ++t) {
      dq += sm.dS[r * AT_PP + t] * sm.f.k[t * AT_PQ + lane];
      dk += sm.dS[t * AT_PP + r] * sm.f.q[t * AT_PQ + lane];
      dv += sm.f.p[t * AT_PP + r] * sm.f.m[t * AT_PP + r] * sm.dO[t * AT_PQ + lane];
    }
    dq *= 0.125f; dk *= 0.125f;
    T* o = obase + (size_t)r * ld + lane;
    if constexpr (sizeof(T) == 2) { o[0] = f2bf(dq); o[H * AT_D] = f2bf(dk); o[2 * H * AT_D] = f2bf(dv); }
    else { o[0] = dq; o[H * AT_D] = dk; o[2 * H * AT_D] = dv; }
  }
}

// ------------------------------------------------------------------------------------------------ attention on MFMA (bf16)
// Same math as the VALU kernels above, on v_mfma_f32_32x32x16_bf16: one wave per (batch, head), the whole 32x32 score tile in
// one accumulator. Forward keeps the KEY index in registers (S^T = K Q^T), so the softmax row reduction is over registers plus
// one cross-half shuffle and the probabilities are directly the B operand of O^T = V^T P^T (accumulator-as-operand with its
// permuted k order; V^T fragments come from ds_read_b64_tr_b16 with the same permutation). Backward keeps the key on the LANE
// (S = Q K^T, dP = dO V^T), so P and dS are directly the B operands of dV^T = dO^T P and dK^T = Q^T dS; only dS crosses LDS once
// (transposed) for dQ^T = K^T dS^T. Dropout masks use the same (seed, site, index) convention as the VALU kernels.
constexpr int AM_PI = 192;   // [32][64] bf16 image pitch (128 B row + 64 B pad)
constexpr int AM_PT = 80;    // [32][32] bf16 image pitch

struct AmFwdSmem { char v[32 * AM_PI]; float madd[32]; };
struct AmBwdSmem { char q[32 * AM_PI], k[32 * AM_PI], d[32 * AM_PI], t[32 * AM_PT]; float m[32 * 33]; float madd[32]; };

DEV void am_load_image(const bf16* base, size_t ld, int L, char* img, int lane) {
  for (int c = lane; c < 256; c += 64) {
    int r = c >> 3, ch = c & 7;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r < L) v = *(const u32x4*)(base + (size_t)r * ld + ch * 8);
    *(u32x4*)(img + r * AM_PI + ch * 16) = v;
  }
}
// row-major fragment straight from global memory: rows x = lane&31 (zero beyond L), k = 16*ks + 8*(lane>>5) + e
DEV bf16x8 am_frag_rows(const bf16* base, size_t ld, int L, int ks, int lane) {
  Chunk16 c;
  c.u = u32x4{0u, 0u, 0u, 0u};
  int r = lane & 31;
  if (r < L) c.u = *(const u32x4*)(base + (size_t)r * ld + ks * 16 + 8 * (lane >> 5));
  return c.h;
}
// transposed fragment from a [k][x] image, natural k order: k = 16*s + 8*(lane>>5) + e
template <int PITCH> DEV bf16x8 am_frag_tr(const char* img, int x0, int s, int lane) {
  int x = x0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  int k = 16 * s + 8 * (lane >> 5) + ((lane >> 2) & 3);
  const char* p = img + k * PITCH + x * 2;
  union { s16x4 v[2]; bf16x8 h; } u;
  u.v[0] = lds_read_tr16(p);
  u.v[1] = lds_read_tr16(p + 4 * PITCH);
  return u.h;
}
// transposed fragment in the k order of an accumulator used as the other operand: element e of lane-half h <-> k = 16s + 8(e>>2) + 4h + (e&3)
template <int PITCH> DEV bf16x8 am_frag_tr_acc(const char* img, int x0, int s, int lane) {
  int x = x0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  int k = 16 * s + 4 * (lane >> 5) + ((lane >> 2) & 3);
  const char* p = img + k * PITCH + x * 2;
  union { s16x4 v[2]; bf16x8 h; } u;
  u.v[0] = lds_read_tr16(p);
  u.v[1] = lds_read_tr16(p + 8 * PITCH);
  return u.h;
}
DEV bf16x8 am_acc_frag(const f32x16& a, int s) {
  bf16x8 f;
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = f2bf(a[8 * s + e]);
  return f;
}
DEV int am_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
// store an accumulator whose rows are d (32 per block) and whose column is a token (lane&31): 4 consecutive d per register group
DEV void am_store_T(bf16* base, size_t ld, int L, int d0, const f32x16& a, float scale, int lane) {
  int tok = lane & 31;
  if (tok >= L) return;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    union { bf16 e[4]; u32x2 u; } pk;
#pragma unroll
    for (int e = 0; e < 4; ++e) pk.e[e] = f2bf(a[4 * g + e] * scale);
    *(u32x2*)(base + (size_t)tok * ld + d0 + 8 * g + 4 * (lane >> 5)) = pk.u;
  }
}

__global__ __launch_bounds__(256) void attention_mfma_fwd_kernel(const bf16* qkv, const int64_t* mask, bf16* ctx, int B, int L, int H, Drop drop) {
  seed_resolve(drop.seed, drop.site);
  __shared__ __attribute__((aligned(16))) AmFwdSmem sm[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int bh = blockIdx.x * 4 + wave;
  const bool live = bh < B * H;
  if (!live) bh = B * H - 1;
  const int b = bh / H, h = bh % H;
  const size_t ld = (size_t)3 * H * 64;
  const bf16* qb = qkv + (size_t)b * L * ld + h * 64;
  AmFwdSmem& S = sm[wave];
  am_load_image(qb + 2 * H * 64, ld, L, S.v, lane);
  if (lane < 32) S.madd[lane] = (lane < L) ? ((mask && mask[(size_t)b * L + lane] == 0) ? AT_NEG : 0.f) : -INFINITY;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)      // S^T[j][i] = sum_d K[j][d] Q[i][d]
    acc = mfma32_bf16(am_frag_rows(qb + H * 64, ld, L, ks, lane), am_frag_rows(qb, ld, L, ks, lane), acc);
  __syncthreads();
  // softmax over j (registers x the two lane halves) for the query i = lane&31
  float mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = acc[r] * 0.125f + S.madd[am_row(r, lane)]; mx = fmaxf(mx, acc[r]); }
  mx = fmaxf(mx, wave_shfl_xor(mx, 32));
  float den = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = __expf(acc[r] - mx); den += acc[r]; }
  den += wave_shfl_xor(den, 32);
  const float inv = 1.0f / den;
  const int i = lane & 31;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float u[4] = {1.f, 1.f, 1.f, 1.f};
    if (drop.p > 0.f) rng_uniform4(drop.seed, drop.site, ((size_t)bh * AT_L + i) * AT_L + 8 * g + 4 * (lane >> 5), u);
    const float ks = drop.p > 0.f ? 1.0f / (1.0f - drop.p) : 1.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[4 * g + e] *= inv * (u[e] >= drop.p ? ks : 0.f);
  }
  // O^T[d][i] = sum_j V^T[d][j] P^T[j][i]
#pragma unroll
  for (int db = 0; db < 2; ++db) {
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) o = mfma32_bf16(am_frag_tr_acc<AM_PI>(S.v, db * 32, s, lane), am_acc_frag(acc, s), o);
    if (live) am_store_T(ctx + (size_t)b * L * ((size_t)H * 64) + h * 64, (size_t)H * 64, L, db * 32, o, 1.f, lane);
  }
}

__global__ __launch_bounds__(128) void attention_mfma_bwd_kernel(const bf16* qkv, const int64_t* mask, const bf16* dctx, bf16* dqkv, int B, int L, int H, Drop drop) {
  seed_resolve(drop.seed, drop.site);
  __shared__ __attribute__((aligned(16))) AmBwdSmem sm[2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int bh = blockIdx.x * 2 + wave;
  const bool live = bh < B * H;
  if (!live) bh = B * H - 1;
  const int b = bh / H, h = bh % H;
  const size_t ld = (size_t)3 * H * 64, ldc = (size_t)H * 64;
  const bf16* qb = qkv + (size_t)b * L * ld + h * 64;
  const bf16* dob = dctx + (size_t)b * L * ldc + h * 64;
  AmBwdSmem& S = sm[wave];
  am_load_image(qb, ld, L, S.q, lane);
  am_load_image(qb + H * 64, ld, L, S.k, lane);
  am_load_image(dob, ldc, L, S.d, lane);
  if (lane < 32) S.madd[lane] = (lane < L) ? ((mask && mask[(size_t)b * L + lane] == 0) ? AT_NEG : 0.f) : -INFINITY;
  {   // dropout multipliers m[i][j]
    const float ks = drop.p > 0.f ? 1.0f / (1.0f - drop.p) : 1.f;
    for (int q4 = lane; q4 < 256; q4 += 64) {
      int i = q4 >> 3, j0 = (q4 & 7) * 4;
      float u[4] = {1.f, 1.f, 1.f, 1.f};
      if (drop.p > 0.f) rng_uniform4(drop.seed, drop.site, ((size_t)bh * AT_L + i) * AT_L + j0, u);
#pragma unroll
      for (int e = 0; e < 4; ++e) S.m[i * 33 + j0 + e] = u[e] >= drop.p ? ks : 0.f;
    }
  }
  f32x16 p, dp;
#pragma unroll
  for (int r = 0; r < 16; ++r) { p[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    p = mfma32_bf16(am_frag_rows(qb, ld, L, ks, lane), am_frag_rows(qb + H * 64, ld, L, ks, lane), p);            // S[i][j]
    dp = mfma32_bf16(am_frag_rows(dob, ldc, L, ks, lane), am_frag_rows(qb + 2 * H * 64, ld, L, ks, lane), dp);    // dP[i][j] = dO V^T
  }
  __syncthreads();
  const int j = lane & 31;
  const float madd = S.madd[j];
  f32x16 pd, ds;
#pragma unroll
  for (int r = 0; r < 16; ++r) {      // softmax over j = across the 32 lanes of a half, per register row i
    float s = p[r] * 0.125f + madd;
    float mx = s;
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) mx = fmaxf(mx, wave_shfl_xor(mx, m));
    float e = __expf(s - mx);
    float den = e;
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) den += wave_shfl_xor(den, m);
    int i = am_row(r, lane);
    float pr = (i < L) ? e / den : 0.f;
    float mul = S.m[i * 33 + j];
    float dpd = dp[r] * mul;
    float t = dpd * pr;
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) t += wave_shfl_xor(t, m);
    p[r] = pr;
    pd[r] = pr * mul;
    ds[r] = pr * (dpd - t);
  }
  // dS^T image for dQ: T[j][i] = dS[i][j]
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    union { bf16 e[4]; u32x2 u; } pk;
#pragma unroll
    for (int e = 0; e < 4; ++e) pk.e[e] = f2bf(ds[4 * g + e]);
    *(u32x2*)(S.t + j * AM_PT + (8 * g + 4 * (lane >> 5)) * 2) = pk.u;
  }
  __syncthreads();
  bf16* ob = dqkv + (size_t)b * L * ld + h * 64;
#pragma unroll
  for (int db = 0; db < 2; ++db) {
    f32x16 dv, dk, dq;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dv[r] = 0.f; dk[r] = 0.f; dq[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      dv = mfma32_bf16(am_frag_tr_acc<AM_PI>(S.d, db * 32, s, lane), am_acc_frag(pd, s), dv);      // dV^T[d][j] = sum_i dO^T[d][i] Pd[i][j]
      dk = mfma32_bf16(am_frag_tr_acc<AM_PI>(S.q, db * 32, s, lane), am_acc_frag(ds, s), dk);      // dK^T[d][j] = sum_i Q^T[d][i] dS[i][j]
      dq = mfma32_bf16(am_frag_tr<AM_PI>(S.k, db * 32, s, lane), am_frag_tr<AM_PT>(S.t, 0, s, lane), dq);   // dQ^T[d][i] = sum_j K^T[d][j] dS^T[j][i]
    }
    if (live) {
      am_store_T(ob + 2 * H * 64, ld, L, db * 32, dv, 1.f, lane);
      am_store_T(ob + H * 64, ld, L, db * 32, dk, 0.125f, lane);
      am_store_T(ob, ld, L, db * 32, dq, 0.125f, lane);
    }
  }
}

// BertPooler backward through tanh: out = dy * (1 - y^2)
template <typename T>
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const T* dy, const T* y, T* out, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    float d[8], v[8];
    load8(dy + i * 8, d);
    load8(y + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] *= 1.f - v[e] * v[e];
    store8(out + i * 8, d);
  }
}

int ew_grid(size_t total) {
  size_t g = (total + 255) / 256;
  return (int)(g < 4096 ? (g ? g : 1) : 4096);
}
int ln_ok(int M, int C) { return M > 0 && C % 8 == 0 && C >= 8 && C <= 64 * LN_MAXCH * 8; }

}  // namespace

#define DISPATCH(dtype, CALL_BF16, CALL_F32) \
  if ((dtype) == CLITE_BF16) { CALL_BF16; } else if ((dtype) == CLITE_F32) { CALL_F32; } else return -1;

extern "C" int clite_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, float eps, void* out, float* stats,
                                   int M, int C, float drop_p, uint64_t drop_seed, uint32_t drop_site, void* stream) {
  if (!ln_ok(M, C) || !x || !out) return -1;
  int grid = (M + 3) / 4;
  if (grid > 2048) grid = 2048;
  Drop d{drop_p, drop_seed, drop_site};
  hipStream_t st = (hipStream_t)stream;
  if (C <= 1024) {
    constexpr int NCHV = 2;
    DISPATCH(dtype,
             hipLaunchKernelGGL((layernorm_fwd_kernel<bf16, NCHV>), dim3(grid), dim3(256), 0, st, (const bf16*)x, gamma, beta, eps, (bf16*)out, stats, M, C, d, LnQ8{}),
             hipLaunchKernelGGL((layernorm_fwd_kernel<float, NCHV>), dim3(grid), dim3(256), 0, st, (const float*)x, gamma, beta, eps, (float*)out, stats, M, C, d, LnQ8{}));
  } else {
    constexpr int NCHV = 4;
    DISPATCH(dtype,
             hipLaunchKernelGGL((layernorm_fwd_kernel<bf16, NCHV>), dim3(grid), dim3(256), 0, st, (const bf16*)x, gamma, beta, eps, (bf16*)out, stats, M, C, d, LnQ8{}),
             hipLaunchKernelGGL((layernorm_fwd_kernel<float, NCHV>), dim3(grid), dim3(256), 0, st, (const float*)x, gamma, beta, eps, (float*)out, stats, M, C, d, LnQ8{}));
  }
  return (int)hipGetLastError();
}

extern "C" int clite_layernorm_fwd_q8(int dtype, const void* x, const float* gamma, const float* beta, float eps, void* out, float* stats,
                                      int M, int C, float drop_p, uint64_t drop_seed, uint32_t drop_site, uint8_t* fp8_out, const float* fp8_scale,
                                      float* fp8_amax, void* stream) {
  if (dtype != CLITE_BF16 || !ln_ok(M, C) || C > 1024 || !x || !out || (fp8_out && !fp8_scale)) return -1;
  int grid = (M + 3) / 4;
  if (grid > 2048) grid = 2048;
  Drop d{drop_p, drop_seed, drop_site};
  hipLaunchKernelGGL((layernorm_fwd_kernel<bf16, 2, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, gamma, beta, eps, (bf16*)out, stats, M, C, d,
                     LnQ8{fp8_out, fp8_scale, fp8_amax});
  return (int)hipGetLastError();
}

extern "C" int clite_layernorm_bwd(int dtype, const void* dy, const void* x, const float* stats, const float* gamma, void* dx, void* dx_masked,
                                   float* dgamma, float* dbeta, float* dcolsum, int M, int C, float in_p, uint64_t in_seed, uint32_t in_site,
                                   float out_p, uint64_t out_seed, uint32_t out_site, void* stream) {
  if (!ln_ok(M, C) || !dy || !x || !stats || !dx) return -1;
  constexpr int NW = 8, LN_BWD_MAX_WG = 256;
  const bool det = clite::deterministic();      // det.h: one workgroup -> one contribution per parameter-gradient address
  int grid = (M + NW - 1) / NW;
  if (grid > LN_BWD_MAX_WG) grid = LN_BWD_MAX_WG;
  if (det) grid = 1;
  Drop di{in_p, in_seed, in_site}, dout{out_p, out_seed, out_site};
  hipStream_t st = (hipStream_t)stream;
  if (C <= 1024) {
    constexpr int NCHV = 2;
    DISPATCH(dtype,
             hipLaunchKernelGGL((layernorm_bwd_kernel<bf16, NCHV, NW>), dim3(grid), dim3(NW * 64), 0, st, (const bf16*)dy, (const bf16*)x, stats, gamma, (bf16*)dx, (bf16*)dx_masked, dgamma, dbeta, dcolsum, M, C, di, dout),
             hipLaunchKernelGGL((layernorm_bwd_kernel<float, NCHV, NW>), dim3(grid), dim3(NW * 64), 0, st, (const float*)dy, (const float*)x, stats, gamma, (float*)dx, (float*)dx_masked, dgamma, dbeta, dcolsum, M, C, di, dout));
  } else {
    constexpr int NCHV = 4, NW4 = 8;      // 4 chunks per lane: 8 waves keep the row state in registers
    grid = (M + NW4 - 1) / NW4;
    if (grid > LN_BWD_MAX_WG) grid = LN_BWD_MAX_WG;
    if (det) grid = 1;
    DISPATCH(dtype,
             hipLaunchKernelGGL((layernorm_bwd_kernel<bf16, NCHV, NW4>), dim3(grid), dim3(NW4 * 64), 0, st, (const bf16*)dy, (const bf16*)x, stats, gamma, (bf16*)dx, (bf16*)dx_masked, dgamma, dbeta, dcolsum, M, C, di, dout),
             hipLaunchKernelGGL((layernorm_bwd_kernel<float, NCHV, NW4>), dim3(grid), dim3(NW4 * 64), 0, st, (const float*)dy, (const float*)x, stats, gamma, (float*)dx, (float*)dx_masked, dgamma, dbeta, dcolsum, M, C, di, dout));
  }
  return (int)hipGetLastError();
}

extern "C" int clite_embed_fwd(int dtype, const int64_t* ids, const void* word, const void* pos, const void* type, void* out,
                               int M, int L, int C, int vocab, void* stream) {
  if (M <= 0 || L <= 0 || C % 8 || !ids || !out) return -1;
  int grid = ew_grid((size_t)M * (C / 8));
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype,
           hipLaunchKernelGGL(embed_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, ids, (const bf16*)word, (const bf16*)pos, (const bf16*)type, (bf16*)out, M, L, C, vocab),
           hipLaunchKernelGGL(embed_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, ids, (const float*)word, (const float*)pos, (const float*)type, (float*)out, M, L, C, vocab));
  return (int)hipGetLastError();
}

extern "C" int clite_embed_bwd(int dtype, const int64_t* ids, const void* d, float* dword, float* dpos, int M, int L, int C, int vocab, int padding_idx,
                               void* stream) {
  if (M <= 0 || L <= 0 || M % L || C % 8 || C > 128 * EMB_COLS || !ids || !d) return -1;
  if (!dword && !dpos) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int egrid = clite::deterministic() ? 1 : L * EMB_SEGS;
  DISPATCH(dtype,
           hipLaunchKernelGGL(embed_bwd_kernel<bf16>, dim3(egrid), dim3(128), 0, st, ids, (const bf16*)d, dword, dpos, M / L, L, C, vocab, padding_idx),
           hipLaunchKernelGGL(embed_bwd_kernel<float>, dim3(egrid), dim3(128), 0, st, ids, (const float*)d, dword, dpos, M / L, L, C, vocab, padding_idx));
  return (int)hipGetLastError();
}

extern "C" int clite_attention_fwd(int dtype, const void* qkv, const int64_t* mask, void* ctx, int B, int L, int H,
                                   float drop_p, uint64_t drop_seed, uint32_t drop_site, void* stream) {
  if (B <= 0 || L <= 0 || L > AT_L || H <= 0 || !qkv || !ctx) return -1;
  Drop d{drop_p, drop_seed, drop_site};
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype,
           hipLaunchKernelGGL(attention_mfma_fwd_kernel, dim3((B * H + 3) / 4), dim3(256), 0, st, (const bf16*)qkv, mask, (bf16*)ctx, B, L, H, d),
           hipLaunchKernelGGL(attention_fwd_kernel<float>, dim3(B * H), dim3(64), 0, st, (const float*)qkv, mask, (float*)ctx, B, L, H, d));
  return (int)hipGetLastError();
}

extern "C" int clite_attention_bwd(int dtype, const void* qkv, const int64_t* mask, const void* dctx, void* dqkv, int B, int L, int H,
                                   float drop_p, uint64_t drop_seed, uint32_t drop_site, void* stream) {
  if (B <= 0 || L <= 0 || L > AT_L || H <= 0 || !qkv || !dctx || !dqkv) return -1;
  Drop d{drop_p, drop_seed, drop_site};
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype,
           hipLaunchKernelGGL(attention_mfma_bwd_kernel, dim3((B * H + 1) / 2), dim3(128), 0, st, (const bf16*)qkv, mask, (const bf16*)dctx, (bf16*)dqkv, B, L, H, d),
           hipLaunchKernelGGL(attention_bwd_kernel<float>, dim3(B * H), dim3(64), 0, st, (const float*)qkv, mask, (const float*)dctx, (float*)dqkv, B, L, H, d));
  return (int)hipGetLastError();
}

extern "C" int clite_tanh_bwd(int dtype, const void* dy, const void* y, void* out, uint64_t n, void* stream) {
  if (!dy || !y || !out || n % 8) return -1;
  int grid = ew_grid(n / 8);
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype,
           hipLaunchKernelGGL(tanh_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)dy, (const bf16*)y, (bf16*)out, (size_t)(n / 8)),
           hipLaunchKernelGGL(tanh_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)dy, (const float*)y, (float*)out, (size_t)(n / 8)));
  return (int)hipGetLastError();
}
