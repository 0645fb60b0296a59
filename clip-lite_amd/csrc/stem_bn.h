// BatchNorm coefficient helpers and the stem's fused max-pool / ReLU / BatchNorm backward gather, shared by resnet_ops.hip (the stand-alone kernels) and
// conv_patch.hip (the stem's fused backward: bn1's backward formed in LDS in front of conv1's weight gradient).
#ifndef CLITE_STEM_BN_H
#define CLITE_STEM_BN_H
#include "intrin.h"
#include "vec.h"
#include "clite.h"

namespace clite {

struct BnCoef { float a[8], b[8]; };

// sum of the R partial accumulators of 8 consecutive channels of one statistic (32-byte vector loads)
DEV void rsum8(const float* p, int replicas, int rstride, float (&s)[8]) {
  load8(p, s);
  for (int r = 1; r < replicas; ++r) {
    float t[8];
    load8(p + (size_t)r * rstride, t);
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] += t[e];
  }
}


// scale/shift of 8 channels from batch statistics (training) or running statistics (eval)
DEV void bn_coef(const float* stats, int R, int RS, const float* gamma, const float* beta, const float* rmean, const float* rvar,
                 int training, int centered, float inv_count, float eps, int C, int c0, BnCoef& k, float (&mean)[8], float (&var)[8]) {
  if (training) {
    // single-pass E[x^2]-E[x]^2 (sums from the conv epilogue), or the two-pass sum of squared deviations when a
    // centered pass (clite_bn_centered_var) filled stats[2][C]
    float s1[8], s2[8];
    rsum8(stats + c0, R, RS, s1);
    rsum8(stats + (centered ? 2 : 1) * C + c0, R, RS, s2);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mean[e] = s1[e] * inv_count;
      var[e] = centered ? s2[e] * inv_count : fmaxf(s2[e] * inv_count - mean[e] * mean[e], 0.f);
    }
  } else {
    load8(rmean + c0, mean);
    load8(rvar + c0, var);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    int c = c0 + e;
    float rstd = rsqrtf(var[e] + eps);
    k.a[e] = gamma[c] * rstd;
    k.b[e] = beta[c];           // applied as (y - mean)*a + b: folding mean into the shift cancels badly when |mean| >> std
  }
}


template <typename T> DEV void round_store_type(float (&v)[8]) { if constexpr (sizeof(T) == 2) round8_bf16(v); }

// dz0 of one input pixel (8 channels): the pooled gradients of the <= 4 windows whose argmax is this pixel, rounded to the storage type,
// masked by relu'(bn(y0)); also returns y0 - mean
template <typename T>
DEV void stem_dz(const T* __restrict__ dpool, const uint8_t* __restrict__ idx, const T* __restrict__ y, int n, int hi, int wi, int H, int W, int Ho, int Wo,
                 int C, int c0, const float (&mean)[8], const BnCoef& k, float (&dz)[8], float (&yc)[8]) {
  // an input pixel lies in at most 2 x 2 pooling windows: odd coordinate -> taps 0 and 2 of windows (c+1)/2 and (c-1)/2, even -> tap 1 of
  // window c/2. All candidates are requested before any is used (clamped address + validity flag); accumulation order r-major, s-minor as in
  // maxpool_bwd.
  int hoc[2], woc[2], rr[2], ss[2];
  bool hv[2], wv[2];
  if (hi & 1) { hoc[0] = (hi + 1) >> 1; rr[0] = 0; hv[0] = hoc[0] < Ho; hoc[1] = (hi - 1) >> 1; rr[1] = 2; hv[1] = true; }
  else        { hoc[0] = hi >> 1;       rr[0] = 1; hv[0] = hoc[0] < Ho; hoc[1] = 0;             rr[1] = 0; hv[1] = false; }
  if (wi & 1) { woc[0] = (wi + 1) >> 1; ss[0] = 0; wv[0] = woc[0] < Wo; woc[1] = (wi - 1) >> 1; ss[1] = 2; wv[1] = true; }
  else        { woc[0] = wi >> 1;       ss[0] = 1; wv[0] = woc[0] < Wo; woc[1] = 0;             ss[1] = 0; wv[1] = false; }
  float d[4][8];
  u32x2 ib[4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int ho = hv[a] ? hoc[a] : 0, wo = wv[b] ? woc[b] : 0;
      const size_t o = (((size_t)n * Ho + ho) * Wo + wo) * C + c0;
      ib[a * 2 + b] = *(const u32x2*)(idx + o);
      load8(dpool + o, d[a * 2 + b]);
    }
  float acc[8];
  zero8(acc);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const uint32_t code = (uint32_t)(rr[a] * 3 + ss[b]);
      const bool valid = hv[a] && wv[b];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t t = ((e < 4 ? ib[a * 2 + b][0] : ib[a * 2 + b][1]) >> (8 * (e & 3))) & 0xFFu;
        if (valid && t == code) acc[e] += d[a * 2 + b][e];
      }
    }
  round_store_type<T>(acc);                          // da0 as maxpool_bwd would have stored it
  float yv[8], a[8];
  load8(y + (((size_t)n * H + hi) * W + wi) * C + c0, yv);
#pragma unroll
  for (int e = 0; e < 8; ++e) { yc[e] = yv[e] - mean[e]; a[e] = relu_f(yc[e] * k.a[e] + k.b[e]); }
  round_store_type<T>(a);
#pragma unroll
  for (int e = 0; e < 8; ++e) dz[e] = a[e] > 0.f ? acc[e] : 0.f;
}

// The same for the 2 x 2 group of input pixels (2 i + dy, 2 j + dx): it lies in exactly the four windows (i + a, j + b), a, b in {0, 1}, which are loaded
// ONCE for the four pixels - nine (pixel, window) pairs instead of the sixteen candidate loads four calls of stem_dz issue (the stand-alone apply
// pass was gather-bound at 55 % of its HBM time). Pixel (dy, dx) takes window (a, b) at tap (dy - 2 a + 1, dx - 2 b + 1) when that tap exists; the
// order of the additions is stem_dz's (the window of the smaller tap row first, then of the smaller tap column), so the sums are bit-identical.
// dz / yc: [dy * 2 + dx][8]; pixels past H / W are computed from clamped addresses and must be dropped by the caller.
template <typename T>
DEV void stem_dz4(const T* __restrict__ dpool, const uint8_t* __restrict__ idx, const T* __restrict__ y, int n, int i, int j, int H, int W, int Ho, int Wo,
                  int C, int c0, const float (&mean)[8], const BnCoef& k, float (&dz)[4][8], float (&yc)[4][8]) {
  float d[4][8];          // window a * 2 + b
  u32x2 ib[4];
  bool wok[4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int ho = i + a, wo = j + b;
      wok[a * 2 + b] = ho < Ho && wo < Wo;
      const size_t o = (((size_t)n * Ho + (ho < Ho ? ho : Ho - 1)) * Wo + (wo < Wo ? wo : Wo - 1)) * C + c0;
      ib[a * 2 + b] = *(const u32x2*)(idx + o);
      load8(dpool + o, d[a * 2 + b]);
    }
  float yv[4][8];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int hi = 2 * i + (q >> 1), wi = 2 * j + (q & 1);
    load8(y + (((size_t)n * H + (hi < H ? hi : H - 1)) * W + (wi < W ? wi : W - 1)) * C + c0, yv[q]);
  }
  auto add = [&](float (&acc)[8], int win, uint32_t code) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t t = ((e < 4 ? ib[win][0] : ib[win][1]) >> (8 * (e & 3))) & 0xFFu;
      if (wok[win] && t == code) acc[e] += d[win][e];
    }
  };
#pragma unroll
  for (int q = 0; q < 4; ++q) zero8(dz[q]);
  // (dy, dx) = (0, 0): window (0, 0) tap (1, 1)
  add(dz[0], 0, 4);
  // (0, 1): window (0, 1) tap (1, 0), then (0, 0) tap (1, 2)
  add(dz[1], 1, 3); add(dz[1], 0, 5);
  // (1, 0): window (1, 0) tap (0, 1), then (0, 0) tap (2, 1)
  add(dz[2], 2, 1); add(dz[2], 0, 7);
  // (1, 1): windows (1, 1) tap (0, 0), (1, 0) tap (0, 2), (0, 1) tap (2, 0), (0, 0) tap (2, 2)
  add(dz[3], 3, 0); add(dz[3], 2, 2); add(dz[3], 1, 6); add(dz[3], 0, 8);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    round_store_type<T>(dz[q]);                      // da0 as maxpool_bwd would have stored it
    float a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { yc[q][e] = yv[q][e] - mean[e]; a[e] = relu_f(yc[q][e] * k.a[e] + k.b[e]); }
    round_store_type<T>(a);
#pragma unroll
    for (int e = 0; e < 8; ++e) dz[q][e] = a[e] > 0.f ? dz[q][e] : 0.f;
  }
}

}  // namespace clite
#endif
