// The MI projection block of the loss heads (reference loss.py:12-40: LN(W2 relu(bn(W1 x)) + b2 + Ws x + bs)): the column-wise and row-wise work
// around its six small GEMMs in four kernels. The window between the encoders' forward and backward is the one place of the step where the chip idles
// (both streams meet at the critic), and what it costs there is dependent launches (~5 us of graph-node latency each), not bytes: ten forward and nine
// backward per block. The GEMMs (128 x 2048 x 2048: 8 MB of weights against 0.5 MB of activations) stay split-K launches of the tile engine that
// accumulate f32 products into a zeroed workspace — a first version of this file multiplied whole 32-column slabs per workgroup straight from global
// memory with no split and lost 0.2 ms per step: every workgroup then reads all of the activation through one CU's L2 port, 40 - 70 us per launch — and
// everything else collapses into:
//   mi_fwd1  per 32-column slab (a workgroup owns its columns for every row of the batch): z = ws[:, :U] stored bf16, BatchNorm1d batch statistics of the
//            stored values, both running-statistics updates, a = relu(bn(z))                                    (was: split-K finish, bn_apply x 2)
//   mi_fwd2  per row: t = ws2 + b2 + (ws[:, U:] + bs) stored bf16, LayerNorm over the row                       (was: two finishes, layernorm_fwd)
//   mi_bwd1  per 32-column slab: da = ws3[:, Fin:] -> mask -> BatchNorm1d backward -> dz; dgamma, dbeta; db2 = dbs = colsum(dtt)   (was: finish,
//            bn_bwd_reduce, bn_bwd_apply, two colsum launches)
//   mi_bwd2  elementwise: dx = ws3[:, :Fin] (+ the prior discriminator's gradient) as bf16                       (was: two finishes)
// No atomics in these kernels: every output element has one writer.
#include "vec.h"
#include "clite.h"

using namespace clite;

namespace {

constexpr int SLAB = 32;          // columns per workgroup of the column-wise kernels
constexpr int RPT = 16;           // rows per thread there: 8 row groups x 16 = the 128-row limit. All of a thread's loads are issued before the first use:
                                  // the operands are L2 / HBM misses of ~2 us each, and a load -> convert -> store loop (stores that may alias the next load)
                                  // is a chain of sixteen of them

// mi_fwd1: one workgroup per 32-column slab of z. ws = f32 [M][2 U]: columns [0, U) = x W1^T, [U, 2 U) = x Ws^T
__global__ __launch_bounds__(256) void mi_fwd1_kernel(clite_mi_block p) {
  __shared__ float red[8][32][2];
  __shared__ float coef[32][2];
  const int tid = threadIdx.x, c = tid & 31, rg = tid >> 5, col = blockIdx.x * SLAB + c;
  const bool cok = col < p.U;
  const int ldw = 2 * p.U;
  const float* __restrict__ ws = p.sc;
  bf16* __restrict__ z = (bf16*)p.z;
  bf16* __restrict__ a = (bf16*)p.a;
  float zv[RPT];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = rg + 8 * i;
    zv[i] = (cok && r < p.M) ? ws[(size_t)r * ldw + col] : 0.f;
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = rg + 8 * i;
    const bf16 zb = f2bf(zv[i]);
    zv[i] = bf2f(zb);
    if (cok && r < p.M) z[(size_t)r * p.U + col] = zb;
    s1 += zv[i]; s2 += zv[i] * zv[i];
  }
  red[rg][c][0] = s1; red[rg][c][1] = s2;
  __syncthreads();
  if (rg == 0) {
    float S1 = 0.f, S2 = 0.f;
    for (int g = 0; g < 8; ++g) { S1 += red[g][c][0]; S2 += red[g][c][1]; }
    const float inv = 1.0f / (float)p.M;
    const float mean = S1 * inv, var = fmaxf(S2 * inv - mean * mean, 0.f);
    coef[c][0] = mean; coef[c][1] = rsqrtf(var + p.eps);
    if (cok) {
      p.stats[col] = S1; p.stats[p.U + col] = S2; p.stats[2 * p.U + col] = 0.f;          // the sums the BatchNorm backward reads (one replica)
      const float unb = p.M > 1 ? (float)p.M / (float)(p.M - 1) : 1.f;
      float rm = p.running_mean[col], rv = p.running_var[col];
      for (int u = 0; u < p.updates; ++u) {          // the reference runs the block twice per step (loss.py:206-222): two updates with the same batch statistics
        rm = (1.f - p.momentum) * rm + p.momentum * mean;
        rv = (1.f - p.momentum) * rv + p.momentum * var * unb;
      }
      p.running_mean[col] = rm; p.running_var[col] = rv;
    }
  }
  __syncthreads();
  if (!cok) return;
  const float mean = coef[c][0], ka = p.gamma[col] * coef[c][1], kb = p.beta[col];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = rg + 8 * i;
    if (r < p.M) a[(size_t)r * p.U + col] = f2bf(relu_f((zv[i] - mean) * ka + kb));
  }
}

// mi_fwd2: one workgroup per row: t = ws2 + b2 + shortcut + bs (stored bf16), then LayerNorm of the stored row (two-pass mean / variance, as clite_layernorm_fwd).
// A thread owns 8 consecutive columns per 2048; U <= 4096
constexpr int F2_IT = 2;
__global__ __launch_bounds__(256) void mi_fwd2_kernel(clite_mi_block p) {
  __shared__ float red[2][4];
  const int tid = threadIdx.x, r = blockIdx.x;
  const float* __restrict__ w2row = p.dxs + (size_t)r * p.U;                 // (forward: dxs carries the second workspace, f32 [M][U] = a W2^T)
  const float* __restrict__ scrow = p.sc + (size_t)r * 2 * p.U + p.U;
  bf16* __restrict__ t = (bf16*)p.t + (size_t)r * p.U;
  bf16* __restrict__ out = (bf16*)p.out + (size_t)r * p.U;
  float v[F2_IT][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < F2_IT; ++i) {
    const int c0 = (tid + 256 * i) * 8;
    if (c0 < p.U) {
      float w[8], b[8];
      load8(w2row + c0, v[i]);
      load8(scrow + c0, w);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] += w[e];
      if (p.b2) { load8(p.b2 + c0, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] += b[e]; }
      if (p.bs) { load8(p.bs + c0, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] += b[e]; }
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[i][e] = bf2f(f2bf(v[i][e])); s += v[i][e]; }
      store8(t + c0, v[i]);
    }
  }
  auto block_sum = [&](float x, int slot) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += wave_shfl_xor(x, m);
    if ((tid & 63) == 0) red[slot][tid >> 6] = x;
    __syncthreads();
    return red[slot][0] + red[slot][1] + red[slot][2] + red[slot][3];
  };
  const float mean = block_sum(s, 0) / (float)p.U;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < F2_IT; ++i)
    if ((tid + 256 * i) * 8 < p.U)
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
  const float rstd = rsqrtf(block_sum(q, 1) / (float)p.U + p.ln_eps);
  if (tid == 0) { p.ln_stats[2 * r] = mean; p.ln_stats[2 * r + 1] = rstd; }
#pragma unroll
  for (int i = 0; i < F2_IT; ++i) {
    const int c0 = (tid + 256 * i) * 8;
    if (c0 < p.U) {
      float g[8], b[8], o[8];
      load8(p.ln_gamma + c0, g);
      load8(p.ln_beta + c0, b);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
      store8(out + c0, o);
    }
  }
}

// mi_bwd1: one workgroup per 32-column slab of da. ws3 = f32 [M][Fin + U]: columns [0, Fin) = dtt Ws (later += dz W1), [Fin, Fin + U) = dtt W2
__global__ __launch_bounds__(256) void mi_bwd1_kernel(clite_mi_block p) {
  __shared__ float red[8][32][3];
  __shared__ float coef[32][2];
  const int tid = threadIdx.x, c = tid & 31, rg = tid >> 5, col = blockIdx.x * SLAB + c;
  const bool cok = col < p.U;
  const int ldw = p.Fin + p.U;
  const bf16* __restrict__ a = (const bf16*)p.a;
  const bf16* __restrict__ z = (const bf16*)p.z;
  const bf16* __restrict__ dtt = (const bf16*)p.dtt;
  const float* __restrict__ ws = p.dxs;
  bf16* __restrict__ dz = (bf16*)p.dz;
  const float inv = 1.0f / (float)p.M;
  float mean = 0.f, rstd = 0.f;
  if (cok) {
    mean = p.stats[col] * inv;
    rstd = rsqrtf(fmaxf(p.stats[p.U + col] * inv - mean * mean, 0.f) + p.eps);
  }
  float d[RPT], zc[RPT];
  float sb = 0.f;
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = rg + 8 * i;
    const bool ok = cok && r < p.M;
    const size_t o = (size_t)r * p.U + col;
    const float av = ok ? bf2f(a[o]) : 0.f;
    const float dv = ok ? ws[(size_t)r * ldw + p.Fin + col] : 0.f;
    zc[i] = ok ? bf2f(z[o]) - mean : 0.f;
    sb += ok ? bf2f(dtt[o]) : 0.f;
    d[i] = av > 0.f ? dv : 0.f;
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < RPT; ++i) { s1 += d[i]; s2 += d[i] * zc[i]; }
  red[rg][c][0] = s1; red[rg][c][1] = s2; red[rg][c][2] = sb;
  __syncthreads();
  if (rg == 0) {
    float S1 = 0.f, S2 = 0.f, SB = 0.f;
    for (int g = 0; g < 8; ++g) { S1 += red[g][c][0]; S2 += red[g][c][1]; SB += red[g][c][2]; }
    coef[c][0] = S1; coef[c][1] = S2;
    if (cok) {
      if (p.dgamma) p.dgamma[col] += rstd * S2;
      if (p.dbeta) p.dbeta[col] += S1;
      if (p.db2) p.db2[col] += SB;
      if (p.dbs) p.dbs[col] += SB;
    }
  }
  __syncthreads();
  if (!cok) return;
  const float S1 = coef[c][0], G = rstd * coef[c][1];
  const float ka = p.gamma[col] * rstd;
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = rg + 8 * i;
    if (r < p.M) dz[(size_t)r * p.U + col] = f2bf(ka * (d[i] - S1 * inv - zc[i] * rstd * G * inv));
  }
}

// mi_bwd2: dx = ws3[:, :Fin] (+ dres), 8 elements per thread
__global__ __launch_bounds__(256) void mi_bwd2_kernel(clite_mi_block p) {
  const int ldw = p.Fin + p.U, per_row = p.Fin / 8;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.M * per_row) return;
  const int r = i / per_row, c0 = (i - r * per_row) * 8;
  float v[8];
  load8(p.dxs + (size_t)r * ldw + c0, v);
  if (p.dres) {
    float d[8];
    load8((const bf16*)p.dres + (size_t)r * p.Fin + c0, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += d[e];
  }
  store8((bf16*)p.dx + (size_t)r * p.Fin + c0, v);
}

bool mi_ok(const clite_mi_block* p) {
  return p && p->M > 0 && p->M <= 8 * RPT && p->Fin > 0 && p->U > 0 && p->U <= 2048 * F2_IT && p->Fin % 16 == 0 && p->U % 16 == 0;
}

}  // namespace

extern "C" int clite_mi_block_fwd1(const clite_mi_block* p, void* stream) {
  if (!mi_ok(p) || !p->sc || !p->gamma || !p->beta || !p->running_mean || !p->running_var || !p->z || !p->a || !p->stats || p->updates < 0) return -1;
  hipLaunchKernelGGL(mi_fwd1_kernel, dim3((p->U + SLAB - 1) / SLAB), dim3(256), 0, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}
extern "C" int clite_mi_block_fwd2(const clite_mi_block* p, void* stream) {
  if (!mi_ok(p) || !p->sc || !p->dxs || !p->t || !p->out || !p->ln_gamma || !p->ln_beta || !p->ln_stats) return -1;
  hipLaunchKernelGGL(mi_fwd2_kernel, dim3(p->M), dim3(256), 0, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}
extern "C" int clite_mi_block_bwd1(const clite_mi_block* p, void* stream) {
  if (!mi_ok(p) || !p->dtt || !p->gamma || !p->z || !p->a || !p->stats || !p->dz || !p->dxs) return -1;
  hipLaunchKernelGGL(mi_bwd1_kernel, dim3((p->U + SLAB - 1) / SLAB), dim3(256), 0, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}
extern "C" int clite_mi_block_bwd2(const clite_mi_block* p, void* stream) {
  if (!mi_ok(p) || !p->dxs || !p->dx || p->Fin % 8) return -1;
  hipLaunchKernelGGL(mi_bwd2_kernel, dim3((p->M * (p->Fin / 8) + 255) / 256), dim3(256), 0, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}
