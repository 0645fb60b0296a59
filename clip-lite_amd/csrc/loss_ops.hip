// JSD mutual-information loss of CLIP-Lite (reference loss.py): the dot-product critic of
// GlobalDiscriminatorDot.forward (loss.py:84-107: L2-normalise both projections, row-wise dot, * exp(temperature)),
// the softplus Jensen-Shannon estimator with roll-by-one negatives (loss.py:206-222,254), the last layer + log terms
// of PriorDiscriminator (loss.py:43-53,188-200) and the total (loss.py:302-305). Reductions are wavefront shuffles;
// one wave owns one sample.
#include "vec.h"
#include "rng.h"
#include "det.h"
#include "clite.h"

using namespace clite;

namespace {

DEV float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // torch F.softplus (beta 1, threshold 20)
DEV float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

constexpr int CR_MAXCH = 4;   // D <= 2048

template <typename T>
DEV void load_row(const T* p, int nchunk, int lane, float (&v)[CR_MAXCH][8]) {
#pragma unroll
  for (int i = 0; i < CR_MAXCH; ++i) {
    int c = lane + 64 * i;
    if (c < nchunk) load8(p + c * 8, v[i]); else zero8(v[i]);
  }
}
DEV float dot_rows(const float (&a)[CR_MAXCH][8], const float (&b)[CR_MAXCH][8]) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CR_MAXCH; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) s += a[i][e] * b[i][e];
  return wave_sum(s);
}

// work[n] = { |a_n|, |b_n|, |b_neg(n)|, cos+, cos-, o+, o-, 0 } ; acc[0] += softplus(-o+)/B ; acc[1] += softplus(o-)/B
// neg: the row of f2 paired with row n as its negative; NULL = roll by one (n + 1 mod B, reference loss.py:214-216). The cluster
// hard-negative branch (loss.py:225-252) passes the permutation [B/2 + i | (i + 1) mod B/2].
template <typename T>
__global__ __launch_bounds__(256) void critic_fwd_kernel(const T* f1, const T* f2, const float* temperature, int B, int D, const int32_t* neg, float* work,
                                                         float* acc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = D / 8;
  const float tscale = expf(temperature[0]);
  for (int n = blockIdx.x * (blockDim.x >> 6) + wave; n < B; n += gridDim.x * (blockDim.x >> 6)) {
    int n1 = neg ? neg[n] : (n + 1 == B ? 0 : n + 1);
    float a[CR_MAXCH][8], b[CR_MAXCH][8], c[CR_MAXCH][8];
    load_row(f1 + (size_t)n * D, nchunk, lane, a);
    load_row(f2 + (size_t)n * D, nchunk, lane, b);
    load_row(f2 + (size_t)n1 * D, nchunk, lane, c);
    float na = fmaxf(sqrtf(dot_rows(a, a)), 1e-12f);
    float nb = fmaxf(sqrtf(dot_rows(b, b)), 1e-12f);
    float nc = fmaxf(sqrtf(dot_rows(c, c)), 1e-12f);
    float cp = dot_rows(a, b) / (na * nb);
    float cn = dot_rows(a, c) / (na * nc);
    float op = cp * tscale, on = cn * tscale;
    if (lane == 0) {
      float* w = work + (size_t)n * 8;
      w[0] = na; w[1] = nb; w[2] = nc; w[3] = cp; w[4] = cn; w[5] = op; w[6] = on; w[7] = 0.f;
      atomic_add_f32(acc + 0, softplus_f(-op) / (float)B);
      atomic_add_f32(acc + 1, softplus_f(on) / (float)B);
    }
  }
}

// out[n] = x[n] / max(|x[n]|_2, 1e-12)  — F.normalize(p=2, dim=-1) of the retrieval path (reference retrieval.py:108,127)
template <typename T>
__global__ __launch_bounds__(256) void l2_normalize_kernel(const T* x, T* out, int B, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = D / 8;
  for (int n = blockIdx.x * (blockDim.x >> 6) + wave; n < B; n += gridDim.x * (blockDim.x >> 6)) {
    float a[CR_MAXCH][8];
    load_row(x + (size_t)n * D, nchunk, lane, a);
    float inv = 1.0f / fmaxf(sqrtf(dot_rows(a, a)), 1e-12f);
#pragma unroll
    for (int i = 0; i < CR_MAXCH; ++i) {
      int c = lane + 64 * i;
      if (c < nchunk) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = a[i][e] * inv;
        store8(out + (size_t)n * D + c * 8, o);
      }
    }
  }
}

// ---- InfoNCE all-pairs variant (BASELINE config 4; not in the reference: symmetric cross-entropy over S = t * C with
// C[i][j] = <a_i, b_j> the cosines of the L2-normalised projections, t = exp(temperature), targets on the diagonal):
//   L = 1/(2B) * [ sum_i (lse_j S[i][j] - S[i][i]) + sum_j (lse_i S[i][j] - S[j][j]) ].
// One wave per line (row: sr = ld, se = 1; column: sr = 1, se = ld); lse[line] kept for backward; acc += sum (lse - S_diag) / (2B).
__global__ __launch_bounds__(256) void infonce_lse_kernel(const float* Cm, int sr, int se, int B, const float* temperature, float* lse, float* acc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float t = expf(temperature[0]);
  for (int i = blockIdx.x * (blockDim.x >> 6) + wave; i < B; i += gridDim.x * (blockDim.x >> 6)) {
    const float* line = Cm + (size_t)i * sr;
    float m = -INFINITY;
    for (int j = lane; j < B; j += 64) m = fmaxf(m, t * line[(size_t)j * se]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < B; j += 64) s += expf(t * line[(size_t)j * se] - m);
    s = wave_sum(s);
    float l = m + logf(s);
    if (lane == 0) {
      lse[i] = l;
      atomic_add_f32(acc, (l - t * line[(size_t)i * se]) / (2.0f * (float)B));
    }
  }
}
// dC[i][j] = g * t * (softmax_row + softmax_col - 2*delta_ij), g = gout*scale/(2B); dtemp += sum_ij g * (...) * S[i][j]  (dS/dtemp = S)
template <typename T>
__global__ __launch_bounds__(256) void infonce_bwd_kernel(const float* Cm, int ld, int B, const float* temperature, const float* lse_r, const float* lse_c,
                                                          const float* gout, float scale, T* dC, int ldd, float* dtemp) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float t = expf(temperature[0]);
  const float g = gout[0] * scale / (2.0f * (float)B);
  float dt = 0.f;
  for (int i = blockIdx.x * (blockDim.x >> 6) + wave; i < B; i += gridDim.x * (blockDim.x >> 6)) {
    const float lr = lse_r[i];
    for (int j0 = lane * 8; j0 < ldd; j0 += 64 * 8) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int j = j0 + e;
        float v = 0.f;
        if (j < B) {
          float sij = t * Cm[(size_t)i * ld + j];
          float d = g * (expf(sij - lr) + expf(sij - lse_c[j]) - (i == j ? 2.f : 0.f));
          dt += d * sij;
          v = d * t;
        }
        o[e] = v;
      }
      store8(dC + (size_t)i * ldd + j0, o);
    }
  }
  dt = wave_sum(dt);
  if (lane == 0 && dtemp) atomic_add_f32(dtemp, dt);
}
// backward of y = x / max(|x|, eps): dx = (dy - y * <dy, y>) / max(|x|, eps)
template <typename T>
__global__ __launch_bounds__(256) void l2_normalize_bwd_kernel(const T* x, const T* y, const T* dy, T* dx, int B, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = D / 8;
  for (int n = blockIdx.x * (blockDim.x >> 6) + wave; n < B; n += gridDim.x * (blockDim.x >> 6)) {
    float a[CR_MAXCH][8], yv[CR_MAXCH][8], d[CR_MAXCH][8];
    load_row(x + (size_t)n * D, nchunk, lane, a);
    load_row(y + (size_t)n * D, nchunk, lane, yv);
    load_row(dy + (size_t)n * D, nchunk, lane, d);
    float inv = 1.0f / fmaxf(sqrtf(dot_rows(a, a)), 1e-12f);
    float p = dot_rows(d, yv);
#pragma unroll
    for (int i = 0; i < CR_MAXCH; ++i) {
      int c = lane + 64 * i;
      if (c < nchunk) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (d[i][e] - yv[i][e] * p) * inv;
        store8(dx + (size_t)n * D + c * 8, o);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void critic_bwd_kernel(const T* f1, const T* f2, const float* temperature, const float* work, const float* gout, float scale,
                                                         int B, int D, const int32_t* neg, const int32_t* neg_inv, T* df1, T* df2, float* dtemp) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = D / 8;
  const float tscale = expf(temperature[0]);
  const float g = gout[0] * scale / (float)B;
  for (int m = blockIdx.x * (blockDim.x >> 6) + wave; m < B; m += gridDim.x * (blockDim.x >> 6)) {
    // m1: the negative partner of row m; mp: the row whose negative partner is m (inverse permutation)
    int m1 = neg ? neg[m] : (m + 1 == B ? 0 : m + 1), mp = neg_inv ? neg_inv[m] : (m == 0 ? B - 1 : m - 1);
    const float* w = work + (size_t)m * 8;
    const float* wp = work + (size_t)mp * 8;
    float na = w[0], nb = w[1], nc = w[2], cp = w[3], cn = w[4];
    float gp = -sigmoid_f(-w[5]) * g, gn = sigmoid_f(w[6]) * g;          // d/do+ , d/do- for sample m
    float gnp = sigmoid_f(wp[6]) * g, cnp = wp[4], nap = wp[0];           // negative pair (m-1, m)
    float a[CR_MAXCH][8], b[CR_MAXCH][8], c[CR_MAXCH][8], ap[CR_MAXCH][8];
    load_row(f1 + (size_t)m * D, nchunk, lane, a);
    load_row(f2 + (size_t)m * D, nchunk, lane, b);
    load_row(f2 + (size_t)m1 * D, nchunk, lane, c);
    load_row(f1 + (size_t)mp * D, nchunk, lane, ap);
#pragma unroll
    for (int i = 0; i < CR_MAXCH; ++i) {
      int ch = lane + 64 * i;
      if (ch < nchunk) {
        float o1[8], o2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float ah = a[i][e] / na, bh = b[i][e] / nb, chh = c[i][e] / nc, aph = ap[i][e] / nap;
          o1[e] = tscale / na * (gp * (bh - cp * ah) + gn * (chh - cn * ah));
          o2[e] = tscale / nb * (gp * (ah - cp * bh) + gnp * (aph - cnp * bh));
        }
        store8(df1 + (size_t)m * D + ch * 8, o1);
        store8(df2 + (size_t)m * D + ch * 8, o2);
      }
    }
    if (lane == 0 && dtemp) atomic_add_f32(dtemp, gp * w[5] + gn * w[6]);
  }
}

// PriorDiscriminator tail on stacked rows [noise u (B rows); features f (B rows)]: logit = h1 . w2 + b2, D = sigmoid(logit)
// acc += -( mean log D(u) + mean log(1 - D(f)) )
template <typename T>
__global__ __launch_bounds__(256) void prior_tail_fwd_kernel(const T* h1, const float* w2, const float* b2, int B, int K, int softplus, float* logit, float* acc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = K / 8;
  for (int r = blockIdx.x * (blockDim.x >> 6) + wave; r < 2 * B; r += gridDim.x * (blockDim.x >> 6)) {
    float s = 0.f;
    for (int c = lane; c < nchunk; c += 64) {
      float h[8], w[8];
      load8(h1 + (size_t)r * K + c * 8, h);
      load8(w2 + c * 8, w);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += h[e] * w[e];
    }
    s = wave_sum(s) + b2[0];
    if (lane == 0) {
      logit[r] = s;
      // -log D = softplus(-s) and -log(1 - D) = softplus(s) mathematically, but not in f32: the reference's prior takes log(sigmoid(s)) and
      // log(1 - sigmoid(s)) (loss.py:190-192), which lose digits for |s| >~ 10, and parity means reproducing that; the `concat` critic's two
      // JSD terms are F.softplus in the reference (loss.py:206-222), selected by `softplus`
      float term;
      if (softplus) {
        term = r < B ? softplus_f(-s) : softplus_f(s);
      } else {
        float d = sigmoid_f(s);
        term = -(r < B ? logf(d) : logf(1.f - d));
      }
      atomic_add_f32(acc, term / (float)B);
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void prior_tail_bwd_kernel(const T* h1, const float* w2, const float* logit, const float* gout, float scale, int B, int K,
                                                             T* dh1, float* dw2, float* db2) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = K / 8;
  const float g = gout[0] * scale / (float)B;
  // dw2 / db2: every row adds to the same K + 1 addresses, and same-address float atomics serialise at the memory side. Each wave keeps its
  // rows' contributions in registers (K <= 512: one 8-column chunk per lane) and adds once at the end; the launch uses few workgroups.
  float aw[8], ab = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) aw[e] = 0.f;
  const bool one_chunk = nchunk <= 64;
  for (int r = blockIdx.x * (blockDim.x >> 6) + wave; r < 2 * B; r += gridDim.x * (blockDim.x >> 6)) {
    float d = sigmoid_f(logit[r]);
    float gl = r < B ? -(1.f - d) * g : d * g;       // d(-log D) = -(1-D), d(-log(1-D)) = D
    for (int c = lane; c < nchunk; c += 64) {
      float h[8], w[8], o[8];
      load8(h1 + (size_t)r * K + c * 8, h);
      load8(w2 + c * 8, w);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o[e] = h[e] > 0.f ? gl * w[e] : 0.f;          // through the ReLU that produced h1
        if (one_chunk) aw[e] += gl * h[e];
        else if (dw2) atomic_add_f32(dw2 + c * 8 + e, gl * h[e]);
      }
      store8(dh1 + (size_t)r * K + c * 8, o);
    }
    ab += gl;
  }
  if (one_chunk && dw2 && lane < nchunk) {
#pragma unroll
    for (int e = 0; e < 8; ++e) atomic_add_f32(dw2 + lane * 8 + e, aw[e]);
  }
  if (lane == 0 && db2) atomic_add_f32(db2, ab);
}

// acc: [-Ej, Em | image prior, text prior | visual -Ej, Em | textual -Ej, Em]
// out[0] = total = (1-w)*(cross + visual + textual) + w*prior (loss.py:302-305), out[1] = cross, out[2] = prior, out[3] = visual, out[4] = textual
__global__ void loss_finalize_kernel(const float* acc, float prior_weight, float* out) {
  if (threadIdx.x == 0) {
    float cross = acc[0] + acc[1], prior = acc[2] + acc[3], visual = acc[4] + acc[5], textual = acc[6] + acc[7];
    out[0] = (1.f - prior_weight) * (cross + visual + textual) + prior_weight * prior;
    out[1] = cross; out[2] = prior; out[3] = visual; out[4] = textual; out[5] = out[6] = out[7] = 0.f;
  }
}

// out = a + b (sums of feature gradients arriving from several loss terms)
template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* a, const T* b, T* out, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    float x[8], y[8];
    load8(a + i * 8, x);
    load8(b + i * 8, y);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] += y[e];
    store8(out + i * 8, x);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void uniform_fill_kernel(T* out, size_t n, uint64_t seed, uint32_t site) {
  seed_resolve(seed, site);
  size_t nch = n / 8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nch; i += (size_t)gridDim.x * 256) {
    float u[8];
    rng_uniform8(seed, site, i * 8, u);
    store8(out + i * 8, u);
  }
}

}  // namespace

// deterministic-reduction mode (det.h): the kernels that add into shared scalars / weight rows run as ONE wave (grid 1 x 64 threads;
// the kernels stride over samples by the waves actually launched), so every sum is formed in sample order
#define DET_SINGLE_WAVE(grid) const bool det_ = clite::deterministic(); if (det_) (grid) = 1
#define DET_BLOCK (det_ ? dim3(64) : dim3(256))
#define DISPATCH(dtype, CALL_BF16, CALL_F32) \
  if ((dtype) == CLITE_BF16) { CALL_BF16; } else if ((dtype) == CLITE_F32) { CALL_F32; } else return -1;

extern "C" int clite_critic_jsd_fwd(int dtype, const void* f1, const void* f2, const float* temperature, int B, int D, const int32_t* neg, float* work,
                                    float* acc, void* stream) {
  if (B <= 0 || D % 8 || D > 64 * CR_MAXCH * 8 || !f1 || !f2 || !work || !acc) return -1;
  int grid = (B + 3) / 4;
  hipStream_t st = (hipStream_t)stream;
  DET_SINGLE_WAVE(grid);
  DISPATCH(dtype,
           hipLaunchKernelGGL(critic_fwd_kernel<bf16>, dim3(grid), DET_BLOCK, 0, st, (const bf16*)f1, (const bf16*)f2, temperature, B, D, neg, work, acc),
           hipLaunchKernelGGL(critic_fwd_kernel<float>, dim3(grid), DET_BLOCK, 0, st, (const float*)f1, (const float*)f2, temperature, B, D, neg, work, acc));
  return (int)hipGetLastError();
}
extern "C" int clite_l2_normalize(int dtype, const void* x, void* out, int B, int D, void* stream) {
  if (B <= 0 || D % 8 || D > 64 * CR_MAXCH * 8 || !x || !out) return -1;
  int grid = (B + 3) / 4;
  if (grid > 2048) grid = 2048;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype,
           hipLaunchKernelGGL(l2_normalize_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)x, (bf16*)out, B, D),
           hipLaunchKernelGGL(l2_normalize_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, (float*)out, B, D));
  return (int)hipGetLastError();
}
extern "C" int clite_infonce_fwd(const float* Cm, int ld, int B, const float* temperature, float* lse_r, float* lse_c, float* acc, void* stream) {
  if (B <= 0 || ld < B || !Cm || !temperature || !lse_r || !lse_c || !acc) return -1;
  int grid = (B + 3) / 4;
  hipStream_t st = (hipStream_t)stream;
  DET_SINGLE_WAVE(grid);
  hipLaunchKernelGGL(infonce_lse_kernel, dim3(grid), DET_BLOCK, 0, st, Cm, ld, 1, B, temperature, lse_r, acc);          // rows: image -> text
  hipLaunchKernelGGL(infonce_lse_kernel, dim3(grid), DET_BLOCK, 0, st, Cm, 1, ld, B, temperature, lse_c, acc + 1);      // columns: text -> image
  return (int)hipGetLastError();
}
extern "C" int clite_infonce_bwd(int dtype, const float* Cm, int ld, int B, const float* temperature, const float* lse_r, const float* lse_c,
                                 const float* gout, float scale, void* dC, int ldd, float* dtemp, void* stream) {
  if (B <= 0 || ld < B || ldd < B || ldd % 8 || !Cm || !temperature || !lse_r || !lse_c || !gout || !dC) return -1;
  int grid = (B + 3) / 4;
  hipStream_t st = (hipStream_t)stream;
  DET_SINGLE_WAVE(grid);
  DISPATCH(dtype,
           hipLaunchKernelGGL(infonce_bwd_kernel<bf16>, dim3(grid), DET_BLOCK, 0, st, Cm, ld, B, temperature, lse_r, lse_c, gout, scale, (bf16*)dC, ldd, dtemp),
           hipLaunchKernelGGL(infonce_bwd_kernel<float>, dim3(grid), DET_BLOCK, 0, st, Cm, ld, B, temperature, lse_r, lse_c, gout, scale, (float*)dC, ldd, dtemp));
  return (int)hipGetLastError();
}
extern "C" int clite_l2_normalize_bwd(int dtype, const void* x, const void* y, const void* dy, void* dx, int B, int D, void* stream) {
  if (B <= 0 || D % 8 || D > 64 * CR_MAXCH * 8 || !x || !y || !dy || !dx) return -1;
  int grid = (B + 3) / 4;
  if (grid > 2048) grid = 2048;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype,
           hipLaunchKernelGGL(l2_normalize_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)x, (const bf16*)y, (const bf16*)dy, (bf16*)dx, B, D),
           hipLaunchKernelGGL(l2_normalize_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, (const float*)y, (const float*)dy, (float*)dx, B, D));
  return (int)hipGetLastError();
}
extern "C" int clite_critic_jsd_bwd(int dtype, const void* f1, const void* f2, const float* temperature, const float* work, const float* gout, float scale,
                                    int B, int D, const int32_t* neg, const int32_t* neg_inv, void* df1, void* df2, float* dtemp, void* stream) {
  if (B <= 0 || D % 8 || D > 64 * CR_MAXCH * 8 || !f1 || !f2 || !work || !gout || !df1 || !df2 || (!neg) != (!neg_inv)) return -1;
  int grid = (B + 3) / 4;
  hipStream_t st = (hipStream_t)stream;
  DET_SINGLE_WAVE(grid);
  DISPATCH(dtype,
           hipLaunchKernelGGL(critic_bwd_kernel<bf16>, dim3(grid), DET_BLOCK, 0, st, (const bf16*)f1, (const bf16*)f2, temperature, work, gout, scale, B, D, neg, neg_inv, (bf16*)df1, (bf16*)df2, dtemp),
           hipLaunchKernelGGL(critic_bwd_kernel<float>, dim3(grid), DET_BLOCK, 0, st, (const float*)f1, (const float*)f2, temperature, work, gout, scale, B, D, neg, neg_inv, (float*)df1, (float*)df2, dtemp));
  return (int)hipGetLastError();
}
extern "C" int clite_prior_tail_fwd(int dtype, const void* h1, const float* w2, const float* b2, int B, int K, int softplus, float* logit, float* acc,
                                    void* stream) {
  if (B <= 0 || K % 8 || !h1 || !w2 || !b2 || !logit || !acc) return -1;
  int grid = (2 * B + 3) / 4;
  hipStream_t st = (hipStream_t)stream;
  DET_SINGLE_WAVE(grid);
  DISPATCH(dtype,
           hipLaunchKernelGGL(prior_tail_fwd_kernel<bf16>, dim3(grid), DET_BLOCK, 0, st, (const bf16*)h1, w2, b2, B, K, softplus, logit, acc),
           hipLaunchKernelGGL(prior_tail_fwd_kernel<float>, dim3(grid), DET_BLOCK, 0, st, (const float*)h1, w2, b2, B, K, softplus, logit, acc));
  return (int)hipGetLastError();
}
extern "C" int clite_prior_tail_bwd(int dtype, const void* h1, const float* w2, const float* logit, const float* gout, float scale, int B, int K,
                                    void* dh1, float* dw2, float* db2, void* stream) {
  if (B <= 0 || K % 8 || !h1 || !w2 || !logit || !gout || !dh1) return -1;
  int grid = (2 * B + 3) / 4;
  if (grid > 16) grid = 16;          // <= 64 waves add into dw2 / db2 (see the kernel)
  hipStream_t st = (hipStream_t)stream;
  DET_SINGLE_WAVE(grid);
  DISPATCH(dtype,
           hipLaunchKernelGGL(prior_tail_bwd_kernel<bf16>, dim3(grid), DET_BLOCK, 0, st, (const bf16*)h1, w2, logit, gout, scale, B, K, (bf16*)dh1, dw2, db2),
           hipLaunchKernelGGL(prior_tail_bwd_kernel<float>, dim3(grid), DET_BLOCK, 0, st, (const float*)h1, w2, logit, gout, scale, B, K, (float*)dh1, dw2, db2));
  return (int)hipGetLastError();
}
extern "C" int clite_loss_finalize(const float* acc, float prior_weight, float* out, void* stream) {
  if (!acc || !out) return -1;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, acc, prior_weight, out);
  return (int)hipGetLastError();
}
extern "C" int clite_add(int dtype, const void* a, const void* b, void* out, uint64_t n, void* stream) {
  if (!a || !b || !out || n % 8) return -1;
  size_t g = (n / 8 + 255) / 256;
  int grid = (int)(g < 2048 ? (g ? g : 1) : 2048);
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype,
           hipLaunchKernelGGL(add_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)a, (const bf16*)b, (bf16*)out, (size_t)(n / 8)),
           hipLaunchKernelGGL(add_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)out, (size_t)(n / 8)));
  return (int)hipGetLastError();
}
extern "C" int clite_uniform_fill(int dtype, void* out, uint64_t n, uint64_t seed, uint32_t site, void* stream) {
  if (!out || n % 8) return -1;
  size_t g = (n / 8 + 255) / 256;
  int grid = (int)(g < 2048 ? (g ? g : 1) : 2048);
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype,
           hipLaunchKernelGGL(uniform_fill_kernel<bf16>, dim3(grid), dim3(256), 0, st, (bf16*)out, (size_t)n, seed, site),
           hipLaunchKernelGGL(uniform_fill_kernel<float>, dim3(grid), dim3(256), 0, st, (float*)out, (size_t)n, seed, site));
  return (int)hipGetLastError();
}
