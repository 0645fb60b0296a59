// HBM-bound kernels of the ResNet image encoder (reference encoder.py:36-65 -> torchvision ResNet; block
// arithmetic restated in-repo at model_zoo/resnet.py:60-100): train-mode BatchNorm2d apply / backward,
// 3x3/2 max-pool, global average pool, NCHW f32 -> padded NHWC image conversion for the 7x7 stem.
// All tensors are NHWC with C % 8 == 0; each lane moves 16-byte (bf16) / 32-byte (f32) vectors; per-channel
// reductions are folded through LDS and leave the workgroup as one float atomic per channel.
#include "vec.h"
#include "det.h"
#include "clite.h"

#include "stem_bn.h"

using namespace clite;

namespace {


// Fold NV per-thread partial sums over all threads of the workgroup that own the same 8-channel chunk (threads tid with equal
// tid % CPR), leaving the total in the threads with tid < CPR (CPR <= 64) or, for CPR > 64, in the threads of wave 0..(CPR/64-1)
// that own distinct chunks. Lanes of one wave that share a chunk are CPR apart -> xor-shuffle tree; the 4 waves meet in LDS.
template <int NV>
DEV bool chunk_fold(float (&v)[NV], int CPR, float* red /* [4][64][NV] */) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int m = 32; m >= CPR && m >= 1; m >>= 1) {
#pragma unroll
    for (int e = 0; e < NV; ++e) v[e] += wave_shfl_xor(v[e], m);
  }
  // after the tree, lanes 0..min(CPR,64)-1 of every wave hold that wave's totals for chunk (wave*64 + lane) % CPR
#pragma unroll
  for (int e = 0; e < NV; ++e) red[(wave * 64 + lane) * NV + e] = v[e];
  __syncthreads();
  if (CPR <= 64) {
    if (tid >= CPR) return false;
#pragma unroll
    for (int e = 0; e < NV; ++e) v[e] = red[tid * NV + e] + red[(64 + tid) * NV + e] + red[(128 + tid) * NV + e] + red[(192 + tid) * NV + e];
    return true;
  }
  // CPR = 128: waves 0,2 and 1,3 own the same chunks; CPR = 256: every thread owns a distinct chunk
  if (CPR == 128) {
    if (tid >= 128) return false;
#pragma unroll
    for (int e = 0; e < NV; ++e) v[e] = red[tid * NV + e] + red[(128 + tid) * NV + e];
    return true;
  }
  return true;
}

// out = relu?( y*a + b  [+ res | + res*a' + b'] ).  Workgroup = 256 threads = (256/CPR) rows x CPR 8-channel chunks.
// NT: streaming (nontemporal) loads and stores for tensors too large to be cache-resident between their producer and consumer
// (>= BN_NT_BYTES per tensor). Stand-alone (tools/probe_bn.py, 205 MB tensors): apply 131 -> 96 us = 6.4 TB/s, backward apply 179 -> 131,
// reduce 108 -> 88, while at <= 51 MB the cached forms are faster (the data still sits in L2 / Infinity Cache). Inside the step, where
// the neighbouring GEMMs compete for the same HBM, the gain shrinks to 1-2 % for the apply kernels and 10 % for the reduce (rocprofv3).
template <bool NT, typename T> DEV void ld8(const T* p, float (&v)[8]) { if constexpr (NT) load8_nt(p, v); else load8(p, v); }
template <bool NT, typename T> DEV void st8(T* p, const float (&v)[8]) { if constexpr (NT) store8_nt(p, v); else store8(p, v); }
constexpr size_t BN_NT_BYTES = (size_t)64 << 20;

// Q (bf16): the producer-fused e4m3 quantiser of the fp8 forward (clite_bn.fp8_out / fp8_scale / fp8_amax): the e4m3 copy of the stored value
// at last step's scale goes out beside it, and this step's max |out| is folded into fp8_amax — one integer atomic max per workgroup.
// S (bf16): clite_bn.out_sum — the column sums of `out` as stored, for the folded BatchNorm backward's weight gradient (bn_fold.hip)
template <typename T, bool NT, bool Q = false, bool S = false>
__global__ __launch_bounds__(256) void bn_apply_kernel(clite_bn p, const T* __restrict__ y, const T* __restrict__ res, T* __restrict__ out, int rows_per_block) {
  const int CPR = p.C / 8, RPS = 256 / CPR;
  const int tid = threadIdx.x, cc = tid % CPR, r0 = tid / CPR, c0 = cc * 8;
  const float inv_count = 1.0f / (float)p.M;
  BnCoef k, kr;
  float mean[8], var[8], mr[8], vr[8];
  bn_coef(p.stats, p.replicas, p.rstride, p.gamma, p.beta, p.running_mean, p.running_var, p.training, p.centered, inv_count, p.eps, p.C, c0, k, mean, var);
  const bool res_affine = res && p.res_gamma;
  if (res_affine)
    bn_coef(p.res_stats, p.replicas, p.rstride, p.res_gamma, p.res_beta, p.res_running_mean, p.res_running_var, p.training, p.centered, inv_count, p.eps, p.C, c0, kr, mr, vr);
  if (blockIdx.x == 0 && r0 == 0 && p.training && p.update_running) {
    float unb = p.M > 1 ? (float)p.M / (float)(p.M - 1) : 1.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      p.running_mean[c0 + e] = (1.f - p.momentum) * p.running_mean[c0 + e] + p.momentum * mean[e];
      p.running_var[c0 + e] = (1.f - p.momentum) * p.running_var[c0 + e] + p.momentum * var[e] * unb;
      if (res_affine) {
        p.res_running_mean[c0 + e] = (1.f - p.momentum) * p.res_running_mean[c0 + e] + p.momentum * mr[e];
        p.res_running_var[c0 + e] = (1.f - p.momentum) * p.res_running_var[c0 + e] + p.momentum * vr[e] * unb;
      }
    }
  }
  int row_begin = blockIdx.x * rows_per_block, row_end = row_begin + rows_per_block;
  if (row_end > p.M) row_end = p.M;
  // four lanes fold their mask bytes into one dword below: that needs the four chunks of a quad in one row (CPR % 4 == 0; C = 8 or 16 keep
  // byte stores) and every lane of a wave in the loop together, hence the wave-uniform trip count with a per-lane `live` predicate
  const bool pack = (CPR & 3) == 0;
  float qmax = 0.f;
  bool qnan = false;
  const float qscale = (Q && p.fp8_out) ? p.fp8_scale[0] : 1.f;
  float osum[8];
  zero8(osum);
#pragma unroll 4
  for (int rb = row_begin; rb < row_end; rb += RPS) {
    const int r = rb + r0;
    const bool live = r < row_end;
    size_t idx = live ? (size_t)r * p.C + c0 : 0;
    float v[8];
    zero8(v);
    if (live) ld8<NT>(y + idx, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (v[e] - mean[e]) * k.a[e] + k.b[e];
    if (res && live) {
      float rv[8];
      load8(res + idx, rv);
      if (res_affine) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (rv[e] - mr[e]) * kr.a[e] + kr.b[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
      }
    }
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = relu_f(v[e]);
      if (p.relu_bits) {
        // the ReLU mask for the backward pass: one byte per 8 channels; consecutive threads own consecutive bytes (byte index = row * C/8 + chunk
        // = a workgroup constant + tid). A 64-lane byte store costs the memory pipeline about what the 16-byte data store next to it costs
        // (bn_apply +4 % in the step), so the four lanes of a quad fold their bytes into one dword (two xor-shuffles) and lane 0 of the quad
        // stores it. Taken from the value as it is stored: (T)v > 0 and v > 0 agree for bf16 / f32 (same exponent range, no flush on conversion).
        uint32_t b = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) b |= (v[e] > 0.f ? 1u : 0u) << e;
        if (pack) {
          b |= (uint32_t)wave_shfl_xor_i((int)b, 1) << 8;
          b |= (uint32_t)wave_shfl_xor_i((int)b, 2) << 16;
          if (live && (tid & 3) == 0) *(uint32_t*)(p.relu_bits + (idx >> 3)) = b;
        } else if (live) {
          p.relu_bits[idx >> 3] = (uint8_t)b;
        }
      }
    }
    if (live) st8<NT>(out + idx, v);
    if constexpr (S) {
      round8_bf16(v);          // the sum of what was stored
#pragma unroll
      for (int e = 0; e < 8; ++e) osum[e] += live ? v[e] : 0.f;
    }
    if constexpr (Q) {
      round8_bf16(v);          // quantise / measure the value as stored, so that the copy equals clite_fp8_quantize of `out` at the same scale
#pragma unroll
      for (int e = 0; e < 8; ++e) { qmax = live ? fmaxf(qmax, fabsf(v[e])) : qmax; qnan = qnan || (live && v[e] != v[e]); }
      if (p.fp8_out && live) {
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // (a NaN passes the clamp untouched and converts to e4m3's NaN, as in fp8_ops.hip)
          const float p0 = v[2 * e] * qscale, q0 = v[2 * e + 1] * qscale;
          const float pp = p0 != p0 ? p0 : fminf(fmaxf(p0, -448.f), 448.f), qq = q0 != q0 ? q0 : fminf(fmaxf(q0, -448.f), 448.f);
          w[e] = cvt2_fp8(pp, qq);
        }
        *(u32x2*)(p.fp8_out + idx) = u32x2{w[0] | (w[1] << 16), w[2] | (w[3] << 16)};
      }
    }
  }
  if constexpr (S) {
    __shared__ float sred[4 * 64 * 8];
    if (chunk_fold<8>(osum, CPR, sred)) {          // one float atomic per column and workgroup, into the workgroup's replica
      float* dst = p.out_sum + (size_t)(blockIdx.x % p.out_sum_replicas) * p.out_sum_stride + c0;
#pragma unroll
      for (int e = 0; e < 8; ++e) atomic_add_f32(dst + e, osum[e]);
    }
  }
  if constexpr (Q) {
    if (p.fp8_amax) {
      __shared__ uint32_t red[4];
      uint32_t mb = qnan ? 0x7FC00000u : f32_bits(qmax);      // non-negative floats order like their bit patterns; the quiet-NaN pattern above all
#pragma unroll
      for (int sh = 32; sh >= 1; sh >>= 1) { const uint32_t o = (uint32_t)wave_shfl_xor_i((int)mb, sh); mb = o > mb ? o : mb; }
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mb;
      __syncthreads();
      if (threadIdx.x == 0) {
        uint32_t b = red[0];
        for (int w = 1; w < 4; ++w) b = red[w] > b ? red[w] : b;
        atomic_max_u32((uint32_t*)p.fp8_amax + (blockIdx.x % CLITE_FP8_AMAX_REPLICAS) * CLITE_FP8_AMAX_STRIDE, b);
      }
    }
  }
}

// dstats[0][c] += sum dz, dstats[1][c] += sum dz*(y - mean_c), with dz = dout * (mask > 0) and mean_c = stats[0][c]/M.
// Centering y here (instead of forming sum dz*y - mean*sum dz afterwards) avoids cancellation on channels with |mean| >> std.
template <typename T, bool NT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dout, const T* __restrict__ mask, const uint8_t* __restrict__ mbits, const T* __restrict__ y, const float* stats, float* dstats, int R, int RS, int M, int C, int rows_per_block) {
  __shared__ float red[256 * 16];
  const int CPR = C / 8, RPS = 256 / CPR;
  const int tid = threadIdx.x, cc = tid % CPR, r0 = tid / CPR, c0 = cc * 8;
  float s1[8], s2[8], mean[8];
  zero8(s1); zero8(s2);
  rsum8(stats + c0, R, RS, mean);
#pragma unroll
  for (int e = 0; e < 8; ++e) mean[e] /= (float)M;
  int row_begin = blockIdx.x * rows_per_block, row_end = row_begin + rows_per_block;
  if (row_end > M) row_end = M;
#pragma unroll 4
  for (int r = row_begin + r0; r < row_end; r += RPS) {
    size_t idx = (size_t)r * C + c0;
    float d[8], yv[8];
    ld8<NT>(dout + idx, d);
    ld8<NT>(y + idx, yv);
    if (mbits) {
      const uint32_t b = mbits[idx >> 3];
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] = (b >> e) & 1u ? d[e] : 0.f;
    } else if (mask) {
      float m[8];
      load8(mask + idx, m);
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] = m[e] > 0.f ? d[e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] += d[e]; s2[e] += d[e] * (yv[e] - mean[e]); }
  }
  float v[16];
#pragma unroll
  for (int e = 0; e < 8; ++e) { v[e] = s1[e]; v[8 + e] = s2[e]; }
  const bool owner = chunk_fold<16>(v, CPR, red);
  __syncthreads();            // everyone is done reading `red`
  if (owner) {                // re-stage as [2][C] so that consecutive lanes add to consecutive addresses (full-rate float atomics)
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[c0 + e] = v[e]; red[C + c0 + e] = v[8 + e]; }
  }
  __syncthreads();
  float* dst = dstats + (size_t)(blockIdx.x % R) * RS;
  for (int i = tid; i < 2 * C; i += 256) atomic_add_f32(dst + i, red[i]);
}

// second pass of a two-pass variance: stats[2][c] += sum (y - mean_c)^2 with mean_c = stats[0][c] / M
template <typename T>
__global__ __launch_bounds__(256) void bn_centered_var_kernel(const T* __restrict__ y, float* stats, int R, int RS, int M, int C, int rows_per_block) {
  __shared__ float red[256 * 8];
  const int CPR = C / 8, RPS = 256 / CPR;
  const int tid = threadIdx.x, cc = tid % CPR, r0 = tid / CPR, c0 = cc * 8;
  float mean[8], s[8];
  const float inv_count = 1.0f / (float)M;
  rsum8(stats + c0, R, RS, mean);
#pragma unroll
  for (int e = 0; e < 8; ++e) { mean[e] *= inv_count; s[e] = 0.f; }
  int row_begin = blockIdx.x * rows_per_block, row_end = row_begin + rows_per_block;
  if (row_end > M) row_end = M;
#pragma unroll 4
  for (int r = row_begin + r0; r < row_end; r += RPS) {
    float v[8];
    load8(y + (size_t)r * C + c0, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) { float d = v[e] - mean[e]; s[e] += d * d; }
  }
  if (chunk_fold<8>(s, CPR, red)) {
    float* dst = stats + (size_t)(blockIdx.x % R) * RS + 2 * C + c0;
#pragma unroll
    for (int e = 0; e < 8; ++e) atomic_add_f32(dst + e, s[e]);
  }
}

// dy = gamma*rstd*(dz - S1/M - xhat*G/M), G = sum dz*xhat = rstd*S2 (S2 = sum dz*(y-mean) from the reduce kernel); dgamma += G, dbeta += S1
// Q (bf16): the producer-fused e5m2 quantiser of the fp8 input gradient that reads dy (clite_bn.fp8_out / fp8_scale / fp8_amax as in bn_apply, with the
// e5m2 format: gradients want its range, the forward's activations e4m3's precision)
template <typename T, bool NT, bool Q = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(clite_bn p, const T* __restrict__ dout, const T* __restrict__ mask, const uint8_t* __restrict__ mbits, const T* __restrict__ y, const float* dstats,
                                                           T* __restrict__ dy, T* __restrict__ dz_out, float* dgamma, float* dbeta, int rows_per_block) {
  const int CPR = p.C / 8, RPS = 256 / CPR;
  const int tid = threadIdx.x, cc = tid % CPR, r0 = tid / CPR, c0 = cc * 8;
  const float inv_count = 1.0f / (float)p.M;
  float mean[8], rstd[8], ka[8], kb[8], kc[8];   // dy = ka*dz + kb + kc*(y - mean)
  float v1[8], v2[8], S1[8], S2[8];
  rsum8(p.stats + c0, p.replicas, p.rstride, v1);
  rsum8(p.stats + (p.centered ? 2 : 1) * p.C + c0, p.replicas, p.rstride, v2);
  rsum8(dstats + c0, p.replicas, p.rstride, S1);
  rsum8(dstats + p.C + c0, p.replicas, p.rstride, S2);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    int c = c0 + e;
    mean[e] = v1[e] * inv_count;
    float var = p.centered ? v2[e] * inv_count : fmaxf(v2[e] * inv_count - mean[e] * mean[e], 0.f);
    rstd[e] = rsqrtf(var + p.eps);
    float G = rstd[e] * S2[e];
    float a = p.gamma[c] * rstd[e];
    // dy = a*(dz - S1/M - (y-mean)*rstd*G/M); (y - mean) is formed explicitly: folding mean into kb cancels badly when |mean| >> std
    ka[e] = a;
    kc[e] = -a * rstd[e] * G * inv_count;
    kb[e] = -a * S1[e] * inv_count;
    if (blockIdx.x == 0 && r0 == 0) {
      if (dgamma) dgamma[c] += G;
      if (dbeta) dbeta[c] += S1[e];
    }
  }
  int row_begin = blockIdx.x * rows_per_block, row_end = row_begin + rows_per_block;
  if (row_end > p.M) row_end = p.M;
  float qmax = 0.f;
  bool qnan = false;
  const float qscale = (Q && p.fp8_out) ? p.fp8_scale[0] : 1.f;
#pragma unroll 4
  for (int r = row_begin + r0; r < row_end; r += RPS) {
    size_t idx = (size_t)r * p.C + c0;
    float d[8], yv[8];
    ld8<NT>(dout + idx, d);
    ld8<NT>(y + idx, yv);
    if (mbits) {
      const uint32_t b = mbits[idx >> 3];
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] = (b >> e) & 1u ? d[e] : 0.f;
    } else if (mask) {
      float m[8];
      load8(mask + idx, m);
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] = m[e] > 0.f ? d[e] : 0.f;
    }
    if (dz_out) store8(dz_out + idx, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] = ka[e] * d[e] + kb[e] + kc[e] * (yv[e] - mean[e]);
    st8<NT>(dy + idx, d);
    if constexpr (Q) {
      round8_bf16(d);          // quantise / measure the value as stored
#pragma unroll
      for (int e = 0; e < 8; ++e) { qmax = fmaxf(qmax, fabsf(d[e])); qnan = qnan || d[e] != d[e]; }
      if (p.fp8_out) {
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // the scale maps max |dy| of an earlier step to 448, 1/128 of e5m2's largest finite value: a gradient that grew since saturates only
          // beyond 128 x that. A NaN passes the clamp untouched and converts to e5m2's NaN.
          const float p0 = d[2 * e] * qscale, q0 = d[2 * e + 1] * qscale;
          const float pp = p0 != p0 ? p0 : fminf(fmaxf(p0, -57344.f), 57344.f), qq = q0 != q0 ? q0 : fminf(fmaxf(q0, -57344.f), 57344.f);
          w[e] = cvt2_bf8(pp, qq);
        }
        *(u32x2*)(p.fp8_out + idx) = u32x2{w[0] | (w[1] << 16), w[2] | (w[3] << 16)};
      }
    }
  }
  if constexpr (Q) {
    if (p.fp8_amax) {
      __shared__ uint32_t red[4];
      uint32_t mb = qnan ? 0x7FC00000u : f32_bits(qmax);
#pragma unroll
      for (int sh = 32; sh >= 1; sh >>= 1) { const uint32_t o = (uint32_t)wave_shfl_xor_i((int)mb, sh); mb = o > mb ? o : mb; }
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mb;
      __syncthreads();
      if (threadIdx.x == 0) {
        uint32_t b = red[0];
        for (int w = 1; w < 4; ++w) b = red[w] > b ? red[w] : b;
        atomic_max_u32((uint32_t*)p.fp8_amax + (blockIdx.x % CLITE_FP8_AMAX_REPLICAS) * CLITE_FP8_AMAX_STRIDE, b);
      }
    }
  }
}

int bn_grid(int M, int C, int* rows_per_block) {
  int CPR = C / 8, RPS = 256 / CPR;
  int sweeps = (M + RPS - 1) / RPS;
  // 16 row sweeps per workgroup amortise the per-channel coefficient prologue on large tensors; small tensors (layer3/4: a few MB)
  // are latency-bound instead and want every CU busy: down to 4 sweeps per workgroup until there are ~1024 workgroups
#ifndef CLITE_BN_MAXWG
#define CLITE_BN_MAXWG 1024
#define CLITE_BN_MINSPW 4
#endif
  int spw = sweeps / CLITE_BN_MAXWG;
  spw = spw < CLITE_BN_MINSPW ? CLITE_BN_MINSPW : (spw > 16 ? 16 : spw);
  int want = (sweeps + spw - 1) / spw;
  int grid = want < 1 ? 1 : (want > CLITE_BN_MAXWG ? CLITE_BN_MAXWG : want);
  int spb = (sweeps + grid - 1) / grid;
  *rows_per_block = spb * RPS;
  return (M + *rows_per_block - 1) / *rows_per_block;
}
// deterministic-reduction mode (det.h): workgroup b adds into replica b % R, so at most R workgroups -> one contribution per address
int bn_grid_reduce(int M, int C, int R, int* rows_per_block) {
  int grid = bn_grid(M, C, rows_per_block);
  if (clite::deterministic() && grid > R) {
    int RPS = 256 / (C / 8);
    int sweeps = (M + RPS - 1) / RPS;
    *rows_per_block = ((sweeps + R - 1) / R) * RPS;
    grid = (M + *rows_per_block - 1) / *rows_per_block;
  }
  return grid;
}
int bn_ok(int M, int C) { return M > 0 && C >= 8 && C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0; }

// ------------------------------------------------------------------------------------------------ pooling
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* x, T* out, uint8_t* idx, int N, int H, int W, int C, int Ho, int Wo) {
  const int CPR = C / 8;
  const uint32_t total = (uint32_t)N * Ho * Wo * CPR;          // 32-bit index arithmetic (the entry point checks the range): 64-bit div/mod
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {          // is emulated and dominated this kernel
    int cc = (int)(i % (uint32_t)CPR);
    uint32_t pix = i / (uint32_t)CPR;
    int wo = (int)(pix % (uint32_t)Wo);
    int ho = (int)((pix / (uint32_t)Wo) % (uint32_t)Ho);
    int n = (int)(pix / ((uint32_t)Wo * Ho));
    float best[8];
    int bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = -1; }
    for (int r = 0; r < 3; ++r)
      for (int s = 0; s < 3; ++s) {
        int hi = ho * 2 - 1 + r, wi = wo * 2 - 1 + s;
        if ((unsigned)hi >= (unsigned)H || (unsigned)wi >= (unsigned)W) continue;
        float v[8];
        load8(x + (((size_t)n * H + hi) * W + wi) * C + cc * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (v[e] > best[e] || bi[e] < 0 || v[e] != v[e]) { best[e] = v[e]; bi[e] = r * 3 + s; }   // first maximum in scan order wins; a NaN wins (as torch)
      }
    store8(out + (size_t)pix * C + cc * 8, best);
    uint8_t* ip = idx + (size_t)pix * C + cc * 8;
    uint32_t lo = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
    uint32_t hi4 = (uint32_t)bi[4] | ((uint32_t)bi[5] << 8) | ((uint32_t)bi[6] << 16) | ((uint32_t)bi[7] << 24);
    *(u32x2*)ip = u32x2{lo, hi4};
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* dout, const uint8_t* idx, T* dx, int N, int H, int W, int C, int Ho, int Wo) {
  const int CPR = C / 8;
  const uint32_t total = (uint32_t)N * H * W * CPR;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    int cc = (int)(i % (uint32_t)CPR);
    uint32_t pix = i / (uint32_t)CPR;
    int wi = (int)(pix % (uint32_t)W);
    int hi = (int)((pix / (uint32_t)W) % (uint32_t)H);
    int n = (int)(pix / ((uint32_t)W * H));
    float acc[8];
    zero8(acc);
    for (int r = 0; r < 3; ++r) {
      int hh = hi + 1 - r;
      if (hh < 0 || (hh & 1)) continue;
      int ho = hh >> 1;
      if (ho >= Ho) continue;
      for (int s = 0; s < 3; ++s) {
        int ww = wi + 1 - s;
        if (ww < 0 || (ww & 1)) continue;
        int wo = ww >> 1;
        if (wo >= Wo) continue;
        size_t o = (((size_t)n * Ho + ho) * Wo + wo) * C + cc * 8;
        u32x2 ib = *(const u32x2*)(idx + o);
        float d[8];
        load8(dout + o, d);
        uint32_t code = (uint32_t)(r * 3 + s);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          uint32_t b = ((e < 4 ? ib[0] : ib[1]) >> (8 * (e & 3))) & 0xFFu;
          if (b == code) acc[e] += d[e];
        }
      }
    }
    store8(dx + (size_t)pix * C + cc * 8, acc);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* x, T* out, int N, int HW, int C) {
  const int CPR = C / 8;
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * CPR) return;
  int n = i / CPR, cc = i % CPR;
  float acc[8];
  zero8(acc);
  for (int p = 0; p < HW; ++p) {
    float v[8];
    load8(x + ((size_t)n * HW + p) * C + cc * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += v[e];
  }
  float inv = 1.0f / (float)HW;
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] *= inv;
  store8(out + (size_t)n * C + cc * 8, acc);
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* dout, T* dx, int N, int HW, int C) {
  const int CPR = C / 8;
  size_t total = (size_t)N * HW * CPR;
  float inv = 1.0f / (float)HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    int cc = (int)(i % CPR);
    size_t pix = i / CPR;
    int n = (int)(pix / HW);
    float v[8];
    load8(dout + (size_t)n * C + cc * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= inv;
    store8(dx + pix * C + cc * 8, v);
  }
}

// ------------------------------------------------------------------------------------------------ fused stem: BN + ReLU + max-pool
// The ResNet stem is conv7x7 -> BatchNorm -> ReLU -> maxpool3x3/2 (torchvision, reference encoder.py:36-38). Its post-BN activation a0 is the
// largest tensor of the step (N x 112 x 112 x 64: 205 MB in bf16 at batch 128) and has exactly one consumer forward (the pool) and one
// backward (the ReLU mask). These three kernels never materialise it, nor the un-pooled gradient da0:
//   forward : pooled = maxpool(relu(bn(y0))) straight from y0                      (reads 205 MB, writes 51 + 26 MB; was 205 r + 205 w + 205 r + 77 w)
//   backward: dz0 = maxpool_bwd(dpool)[pixel] * (bn(y0) > 0) is re-formed per input pixel from dpool / idx / y0, once for the two BatchNorm
//             reductions and once for dy0                                          (282 + 487 MB instead of 282 + 615 + 820 MB)
// Values are rounded to the storage type exactly where the unfused kernels stored them (a0 before the max, da0 before the mask), so the
// results are bit-identical to bn_apply -> maxpool / maxpool_bwd -> bn_bwd_reduce -> bn_bwd_apply.

template <typename T>
__global__ __launch_bounds__(256) void stem_bn_pool_fwd_kernel(clite_bn p, const T* __restrict__ y, T* __restrict__ out, uint8_t* __restrict__ idx, T* __restrict__ ymax,
                                                               int N, int H, int W, int Ho, int Wo, int rows_per_block) {
  const int CPR = p.C / 8, RPS = 256 / CPR;
  const int tid = threadIdx.x, cc = tid % CPR, r0 = tid / CPR, c0 = cc * 8;
  BnCoef k;
  float mean[8], var[8];
  bn_coef(p.stats, p.replicas, p.rstride, p.gamma, p.beta, p.running_mean, p.running_var, p.training, p.centered, 1.0f / (float)p.M, p.eps, p.C, c0, k, mean, var);
  if (blockIdx.x == 0 && r0 == 0 && p.training && p.update_running) {
    float unb = p.M > 1 ? (float)p.M / (float)(p.M - 1) : 1.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      p.running_mean[c0 + e] = (1.f - p.momentum) * p.running_mean[c0 + e] + p.momentum * mean[e];
      p.running_var[c0 + e] = (1.f - p.momentum) * p.running_var[c0 + e] + p.momentum * var[e] * unb;
    }
  }
  const int P = N * Ho * Wo;
  int row_begin = blockIdx.x * rows_per_block, row_end = row_begin + rows_per_block;
  if (row_end > P) row_end = P;
  for (int pix = row_begin + r0; pix < row_end; pix += RPS) {
    const int wo = pix % Wo, ho = (pix / Wo) % Ho, n = pix / (Wo * Ho);
    // all nine taps are requested before any is used (clamped address + validity flag instead of a branch around the load)
    float tap[9][8];
    bool ok[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int hi = ho * 2 - 1 + r, wi = wo * 2 - 1 + s;
        ok[r * 3 + s] = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
        const int hc = hi < 0 ? 0 : (hi >= H ? H - 1 : hi), wc = wi < 0 ? 0 : (wi >= W ? W - 1 : wi);
        load8(y + (((size_t)n * H + hc) * W + wc) * p.C + c0, tap[r * 3 + s]);
      }
    float best[8], ybest[8];
    int bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = -1; ybest[e] = 0.f; }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = relu_f((tap[t][e] - mean[e]) * k.a[e] + k.b[e]);
      round_store_type<T>(v);                      // a0 as bn_apply would have stored it
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (ok[t] && (v[e] > best[e] || bi[e] < 0 || v[e] != v[e])) { best[e] = v[e]; bi[e] = t; ybest[e] = tap[t][e]; }   // first maximum in scan order wins; a NaN wins (torch max_pool2d propagates it)
    }
    store8(out + (size_t)pix * p.C + c0, best);
    // clite_stem_bn_pool_fwd_ex: the BatchNorm INPUT at the argmax (an exact copy of that y element) and the packed relu' bits of the pooled output -
    // what lets the kernel that writes the pooled gradient accumulate this BatchNorm's two backward reductions in its epilogue (clite_epilogue.bn_y
    // := ymax, relu_bits), instead of a pass over the 4 x larger un-pooled tensors
    if (ymax) store8(ymax + (size_t)pix * p.C + c0, ybest);
    if (p.relu_bits) {
      uint32_t b = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) b |= (best[e] > 0.f ? 1u : 0u) << e;
      p.relu_bits[((size_t)pix * p.C + c0) >> 3] = (uint8_t)b;
    }
    uint32_t lo = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
    uint32_t hi4 = (uint32_t)bi[4] | ((uint32_t)bi[5] << 8) | ((uint32_t)bi[6] << 16) | ((uint32_t)bi[7] << 24);
    *(u32x2*)(idx + (size_t)pix * p.C + c0) = u32x2{lo, hi4};
  }
}

template <typename T>
__global__ __launch_bounds__(256) void stem_bn_pool_bwd_reduce_kernel(clite_bn p, const T* __restrict__ dpool, const uint8_t* __restrict__ idx, const T* __restrict__ y,
                                                                      float* dstats, int N, int H, int W, int Ho, int Wo, int rows_per_block) {
  __shared__ float red[256 * 16];
  const int C = p.C, CPR = C / 8, RPS = 256 / CPR;
  const int tid = threadIdx.x, cc = tid % CPR, r0 = tid / CPR, c0 = cc * 8;
  BnCoef k;
  float mean[8], var[8], s1[8], s2[8];
  bn_coef(p.stats, p.replicas, p.rstride, p.gamma, p.beta, p.running_mean, p.running_var, 1, p.centered, 1.0f / (float)p.M, p.eps, C, c0, k, mean, var);
  zero8(s1); zero8(s2);
  const int Q = N * H * W;
  int row_begin = blockIdx.x * rows_per_block, row_end = row_begin + rows_per_block;
  if (row_end > Q) row_end = Q;
  for (int q = row_begin + r0; q < row_end; q += RPS) {
    const int wi = q % W, hi = (q / W) % H, n = q / (W * H);
    float dz[8], yc[8];
    stem_dz(dpool, idx, y, n, hi, wi, H, W, Ho, Wo, C, c0, mean, k, dz, yc);
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] += dz[e]; s2[e] += dz[e] * yc[e]; }
  }
  float v[16];
#pragma unroll
  for (int e = 0; e < 8; ++e) { v[e] = s1[e]; v[8 + e] = s2[e]; }
  const bool owner = chunk_fold<16>(v, CPR, red);
  __syncthreads();
  if (owner) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[c0 + e] = v[e]; red[C + c0 + e] = v[8 + e]; }
  }
  __syncthreads();
  float* dst = dstats + (size_t)(blockIdx.x % p.replicas) * p.rstride;
  for (int i = tid; i < 2 * C; i += 256) atomic_add_f32(dst + i, red[i]);
}

template <typename T>
__global__ __launch_bounds__(256) void stem_bn_pool_bwd_apply_kernel(clite_bn p, const T* __restrict__ dpool, const uint8_t* __restrict__ idx, const T* __restrict__ y,
                                                                     const float* dstats, T* __restrict__ dy, float* dgamma, float* dbeta,
                                                                     int N, int H, int W, int Ho, int Wo, int rows_per_block) {
  const int C = p.C, CPR = C / 8, RPS = 256 / CPR;
  const int tid = threadIdx.x, cc = tid % CPR, r0 = tid / CPR, c0 = cc * 8;
  const float inv_count = 1.0f / (float)p.M;
  BnCoef k;
  float mean[8], var[8], ka[8], kb[8], kc[8], S1[8], S2[8];
  bn_coef(p.stats, p.replicas, p.rstride, p.gamma, p.beta, p.running_mean, p.running_var, 1, p.centered, inv_count, p.eps, C, c0, k, mean, var);
  rsum8(dstats + c0, p.replicas, p.rstride, S1);
  rsum8(dstats + C + c0, p.replicas, p.rstride, S2);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c0 + e;
    const float rstd = rsqrtf(var[e] + p.eps);
    const float G = rstd * S2[e];
    const float a = p.gamma[c] * rstd;
    ka[e] = a;
    kc[e] = -a * rstd * G * inv_count;
    kb[e] = -a * S1[e] * inv_count;
    if (blockIdx.x == 0 && r0 == 0) {
      if (dgamma) dgamma[c] += G;
      if (dbeta) dbeta[c] += S1[e];
    }
  }
  const int Q = N * H * W;
  int row_begin = blockIdx.x * rows_per_block, row_end = row_begin + rows_per_block;
  if (row_end > Q) row_end = Q;
  for (int q = row_begin + r0; q < row_end; q += RPS) {
    const int wi = q % W, hi = (q / W) % H, n = q / (W * H);
    float dz[8], yc[8];
    stem_dz(dpool, idx, y, n, hi, wi, H, W, Ho, Wo, C, c0, mean, k, dz, yc);
#pragma unroll
    for (int e = 0; e < 8; ++e) dz[e] = ka[e] * dz[e] + kb[e] + kc[e] * yc[e];
    store8(dy + (size_t)q * C + c0, dz);
  }
}

// The apply pass over 2 x 2 pixel groups (stem_bn.h: stem_dz4): a thread owns the four pixels of a group and 8 channels; four window loads serve them.
template <typename T>
__global__ __launch_bounds__(256) void stem_bn_pool_bwd_apply4_kernel(clite_bn p, const T* __restrict__ dpool, const uint8_t* __restrict__ idx, const T* __restrict__ y,
                                                                      const float* dstats, T* __restrict__ dy, float* dgamma, float* dbeta,
                                                                      int N, int H, int W, int Ho, int Wo, int groups_per_block) {
  const int C = p.C, CPR = C / 8, RPS = 256 / CPR;
  const int tid = threadIdx.x, cc = tid % CPR, r0 = tid / CPR, c0 = cc * 8;
  const float inv_count = 1.0f / (float)p.M;
  BnCoef k;
  float mean[8], var[8], ka[8], kb[8], kc[8], S1[8], S2[8];
  bn_coef(p.stats, p.replicas, p.rstride, p.gamma, p.beta, p.running_mean, p.running_var, 1, p.centered, inv_count, p.eps, C, c0, k, mean, var);
  rsum8(dstats + c0, p.replicas, p.rstride, S1);
  rsum8(dstats + C + c0, p.replicas, p.rstride, S2);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c0 + e;
    const float rstd = rsqrtf(var[e] + p.eps);
    const float G = rstd * S2[e];
    const float a = p.gamma[c] * rstd;
    ka[e] = a;
    kc[e] = -a * rstd * G * inv_count;
    kb[e] = -a * S1[e] * inv_count;
    if (blockIdx.x == 0 && r0 == 0) {
      if (dgamma) dgamma[c] += G;
      if (dbeta) dbeta[c] += S1[e];
    }
  }
  const int H2 = (H + 1) >> 1, W2 = (W + 1) >> 1;
  const int Q = N * H2 * W2;
  int g_begin = blockIdx.x * groups_per_block, g_end = g_begin + groups_per_block;
  if (g_end > Q) g_end = Q;
  for (int g = g_begin + r0; g < g_end; g += RPS) {
    const int j = g % W2, i = (g / W2) % H2, n = g / (W2 * H2);
    float dz[4][8], yc[4][8];
    stem_dz4(dpool, idx, y, n, i, j, H, W, Ho, Wo, C, c0, mean, k, dz, yc);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int hi = 2 * i + (q >> 1), wi = 2 * j + (q & 1);
      if (hi >= H || wi >= W) continue;
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = ka[e] * dz[q][e] + kb[e] + kc[e] * yc[q][e];
      store8(dy + (((size_t)n * H + hi) * W + wi) * C + c0, o);
    }
  }
}

// image f32 NCHW [N][3][H][W] -> T [N][H+2*pad][Wp][4], zero padded (channel 3 = 0). One wave per output row (n, hp), a lane per group of 4 INPUT
// columns: three 16-byte loads (one per channel plane) and four 8-byte stores, 32-bit index arithmetic and one division per wave (round 3's form
// spent its 125 us on three 64-bit divisions per pixel; this one streams at the HBM rate).
template <typename T>
__global__ __launch_bounds__(256) void image_to_nhwc4_kernel(const float* __restrict__ img, T* __restrict__ out, int N, int H, int W, int pad, int Hp, int Wp) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);          // (n, hp)
  if (row >= N * Hp) return;
  const int lane = threadIdx.x & 63;
  const int n = row / Hp, hp = row - n * Hp, h = hp - pad;
  const bool hv = (unsigned)h < (unsigned)H;
  const size_t plane = (size_t)H * W;
  const float* src = img + (size_t)n * 3 * plane + (size_t)(hv ? h : 0) * W;
  T* dst = out + (size_t)row * Wp * 4;
  const int g0 = -((pad + 3) >> 2);                             // first group of 4 input columns that reaches output column 0
  const bool vec = (W & 3) == 0 && (((uintptr_t)img) & 15) == 0;
  for (int g = g0 + lane; g * 4 + pad < Wp; g += 64) {
    const int w0 = g * 4;
    float v[3][4];
    if (hv && vec && w0 >= 0 && w0 + 3 < W) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const f32x4 q = *(const f32x4*)(src + c * plane + w0);
        v[c][0] = q[0]; v[c][1] = q[1]; v[c][2] = q[2]; v[c][3] = q[3];
      }
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[c][e] = (hv && (unsigned)(w0 + e) < (unsigned)W) ? src[c * plane + w0 + e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int wp = w0 + e + pad;
      if ((unsigned)wp >= (unsigned)Wp) continue;
      if constexpr (sizeof(T) == 2) {
        union { bf16 x[4]; u32x2 u; } pk;
        pk.x[0] = f2bf(v[0][e]); pk.x[1] = f2bf(v[1][e]); pk.x[2] = f2bf(v[2][e]); pk.x[3] = f2bf(0.f);
        *(u32x2*)(dst + (size_t)wp * 4) = pk.u;
      } else {
        *(f32x4*)(dst + (size_t)wp * 4) = f32x4{v[0][e], v[1][e], v[2][e], 0.f};
      }
    }
  }
}

// out[c] += sum_m x[m][c]  (bias gradients). Workgroup = 32 column chunks (256 columns) x 8 row lanes; grid = (column
// groups, row slabs); the 8 row lanes fold through LDS and every slab ends in one float atomic per column.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* x, float* out, int M, int N, int rows_per_block, int nfold) {
  __shared__ float red[8][32 * 8];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int chunk = blockIdx.x * 32 + cx;
  const bool cvalid = chunk < N / 8;
  int row_begin = blockIdx.y * rows_per_block, row_end = row_begin + rows_per_block;
  if (row_end > M) row_end = M;
  float acc[8];
  zero8(acc);
  if (cvalid) {
#pragma unroll 4
    for (int r = row_begin + ry; r < row_end; r += 8) {
      float v[8];
      load8(x + (size_t)r * N + chunk * 8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[ry][cx * 8 + e] = acc[e];
  __syncthreads();
  int col = threadIdx.x;                       // 256 columns of this group
  int gcol = blockIdx.x * 256 + col;
  if (gcol < N) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += red[r][col];
    atomic_add_f32(out + (nfold ? gcol % nfold : gcol), s);          // nfold: the matrix is a narrow [M * N / nfold][nfold] one seen 256 columns wide (clite_colsum)
  }
}

int ew_grid(size_t total) {
  size_t g = (total + 255) / 256;
  return (int)(g < 4096 ? (g ? g : 1) : 4096);
}

}  // namespace

#define DISPATCH(dtype, CALL_BF16, CALL_F32) \
  if ((dtype) == CLITE_BF16) { CALL_BF16; } else if ((dtype) == CLITE_F32) { CALL_F32; } else return -1;

extern "C" int clite_bn_apply(const clite_bn* p, int dtype, const void* y, const void* res, void* out, void* stream) {
  if (!p || !bn_ok(p->M, p->C) || !y || !out || p->replicas < 1) return -1;
  int rpb;
  int grid = bn_grid(p->M, p->C, &rpb);
  hipStream_t st = (hipStream_t)stream;
  const bool nt = (size_t)p->M * p->C * (dtype == CLITE_BF16 ? 2 : 4) >= BN_NT_BYTES;
  if (p->out_sum) {          // column sums of the stored output (clite_bn.out_sum; bf16, not together with the fp8 producer)
    if (dtype != CLITE_BF16 || p->fp8_out || p->fp8_amax || p->out_sum_replicas < 1) return -1;
    if (nt) hipLaunchKernelGGL((bn_apply_kernel<bf16, true, false, true>), dim3(grid), dim3(256), 0, st, *p, (const bf16*)y, (const bf16*)res, (bf16*)out, rpb);
    else hipLaunchKernelGGL((bn_apply_kernel<bf16, false, false, true>), dim3(grid), dim3(256), 0, st, *p, (const bf16*)y, (const bf16*)res, (bf16*)out, rpb);
    return (int)hipGetLastError();
  }
  if (p->fp8_out || p->fp8_amax) {          // producer-fused e4m3 copy / amax (bf16 activations only)
    if (dtype != CLITE_BF16 || (p->fp8_out && !p->fp8_scale)) return -1;
    if (nt) hipLaunchKernelGGL((bn_apply_kernel<bf16, true, true>), dim3(grid), dim3(256), 0, st, *p, (const bf16*)y, (const bf16*)res, (bf16*)out, rpb);
    else hipLaunchKernelGGL((bn_apply_kernel<bf16, false, true>), dim3(grid), dim3(256), 0, st, *p, (const bf16*)y, (const bf16*)res, (bf16*)out, rpb);
    return (int)hipGetLastError();
  }
  if (nt) {
    DISPATCH(dtype,
             hipLaunchKernelGGL((bn_apply_kernel<bf16, true>), dim3(grid), dim3(256), 0, st, *p, (const bf16*)y, (const bf16*)res, (bf16*)out, rpb),
             hipLaunchKernelGGL((bn_apply_kernel<float, true>), dim3(grid), dim3(256), 0, st, *p, (const float*)y, (const float*)res, (float*)out, rpb));
  } else {
    DISPATCH(dtype,
             hipLaunchKernelGGL((bn_apply_kernel<bf16, false>), dim3(grid), dim3(256), 0, st, *p, (const bf16*)y, (const bf16*)res, (bf16*)out, rpb),
             hipLaunchKernelGGL((bn_apply_kernel<float, false>), dim3(grid), dim3(256), 0, st, *p, (const float*)y, (const float*)res, (float*)out, rpb));
  }
  return (int)hipGetLastError();
}

extern "C" int clite_bn_centered_var(int dtype, const void* y, float* stats, int replicas, int rstride, int M, int C, void* stream) {
  if (!bn_ok(M, C) || !y || !stats || replicas < 1) return -1;
  int rpb;
  int grid = bn_grid_reduce(M, C, replicas, &rpb);
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype,
           hipLaunchKernelGGL(bn_centered_var_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)y, stats, replicas, rstride, M, C, rpb),
           hipLaunchKernelGGL(bn_centered_var_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)y, stats, replicas, rstride, M, C, rpb));
  return (int)hipGetLastError();
}

extern "C" int clite_bn_bwd_reduce(int dtype, const void* dout, const void* mask, const uint8_t* mask_bits, const void* y, const float* stats, float* dstats,
                                   int replicas, int rstride, int M, int C, void* stream) {
  if (!bn_ok(M, C) || !dout || !y || !stats || !dstats || replicas < 1 || (mask && mask_bits)) return -1;
  int rpb;
  int grid = bn_grid_reduce(M, C, replicas, &rpb);
  hipStream_t st = (hipStream_t)stream;
  const bool nt = (size_t)M * C * (dtype == CLITE_BF16 ? 2 : 4) >= BN_NT_BYTES;
  if (nt) {
    DISPATCH(dtype,
             hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16, true>), dim3(grid), dim3(256), 0, st, (const bf16*)dout, (const bf16*)mask, mask_bits, (const bf16*)y, stats, dstats, replicas, rstride, M, C, rpb),
             hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, true>), dim3(grid), dim3(256), 0, st, (const float*)dout, (const float*)mask, mask_bits, (const float*)y, stats, dstats, replicas, rstride, M, C, rpb));
  } else {
    DISPATCH(dtype,
             hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16, false>), dim3(grid), dim3(256), 0, st, (const bf16*)dout, (const bf16*)mask, mask_bits, (const bf16*)y, stats, dstats, replicas, rstride, M, C, rpb),
             hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, false>), dim3(grid), dim3(256), 0, st, (const float*)dout, (const float*)mask, mask_bits, (const float*)y, stats, dstats, replicas, rstride, M, C, rpb));
  }
  return (int)hipGetLastError();
}

extern "C" int clite_bn_bwd_apply(const clite_bn* p, int dtype, const void* dout, const void* mask, const uint8_t* mask_bits, const void* y, const float* dstats,
                                  void* dy, void* dz, float* dgamma, float* dbeta, void* stream) {
  if (!p || !bn_ok(p->M, p->C) || !dout || !y || !dstats || !dy || p->replicas < 1 || (mask && mask_bits)) return -1;
  int rpb;
  int grid = bn_grid(p->M, p->C, &rpb);
  hipStream_t st = (hipStream_t)stream;
  const bool nt = (size_t)p->M * p->C * (dtype == CLITE_BF16 ? 2 : 4) >= BN_NT_BYTES;
  if (p->fp8_out || p->fp8_amax) {          // producer-fused e5m2 copy / amax of dy (bf16 only)
    if (dtype != CLITE_BF16 || (p->fp8_out && !p->fp8_scale)) return -1;
    if (nt) hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16, true, true>), dim3(grid), dim3(256), 0, st, *p, (const bf16*)dout, (const bf16*)mask, mask_bits, (const bf16*)y, dstats, (bf16*)dy, (bf16*)dz, dgamma, dbeta, rpb);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16, false, true>), dim3(grid), dim3(256), 0, st, *p, (const bf16*)dout, (const bf16*)mask, mask_bits, (const bf16*)y, dstats, (bf16*)dy, (bf16*)dz, dgamma, dbeta, rpb);
    return (int)hipGetLastError();
  }
  if (nt) {
    DISPATCH(dtype,
             hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16, true>), dim3(grid), dim3(256), 0, st, *p, (const bf16*)dout, (const bf16*)mask, mask_bits, (const bf16*)y, dstats, (bf16*)dy, (bf16*)dz, dgamma, dbeta, rpb),
             hipLaunchKernelGGL((bn_bwd_apply_kernel<float, true>), dim3(grid), dim3(256), 0, st, *p, (const float*)dout, (const float*)mask, mask_bits, (const float*)y, dstats, (float*)dy, (float*)dz, dgamma, dbeta, rpb));
  } else {
    DISPATCH(dtype,
             hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16, false>), dim3(grid), dim3(256), 0, st, *p, (const bf16*)dout, (const bf16*)mask, mask_bits, (const bf16*)y, dstats, (bf16*)dy, (bf16*)dz, dgamma, dbeta, rpb),
             hipLaunchKernelGGL((bn_bwd_apply_kernel<float, false>), dim3(grid), dim3(256), 0, st, *p, (const float*)dout, (const float*)mask, mask_bits, (const float*)y, dstats, (float*)dy, (float*)dz, dgamma, dbeta, rpb));
  }
  return (int)hipGetLastError();
}

extern "C" int clite_maxpool3x3s2_fwd(int dtype, const void* x, void* out, uint8_t* idx, int N, int H, int W, int C, void* stream) {
  if (C % 8 || N <= 0 || (size_t)N * H * W * (C / 8) >= ((size_t)1 << 31)) return -1;
  int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  int grid = ew_grid((size_t)N * Ho * Wo * (C / 8));
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype,
           hipLaunchKernelGGL(maxpool_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)x, (bf16*)out, idx, N, H, W, C, Ho, Wo),
           hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, (float*)out, idx, N, H, W, C, Ho, Wo));
  return (int)hipGetLastError();
}

extern "C" int clite_maxpool3x3s2_bwd(int dtype, const void* dout, const uint8_t* idx, void* dx, int N, int H, int W, int C, void* stream) {
  if (C % 8 || N <= 0 || (size_t)N * H * W * (C / 8) >= ((size_t)1 << 31)) return -1;
  int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  int grid = ew_grid((size_t)N * H * W * (C / 8));
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype,
           hipLaunchKernelGGL(maxpool_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)dout, idx, (bf16*)dx, N, H, W, C, Ho, Wo),
           hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)dout, idx, (float*)dx, N, H, W, C, Ho, Wo));
  return (int)hipGetLastError();
}

// the stem's apply pass over 2 x 2 pixel groups (both entry points that run it)
static int stem_apply4(const clite_bn& p, int dtype, const void* dpool, const uint8_t* idx, const void* y, const float* dstats, void* dy, float* dgamma, float* dbeta,
                       int N, int H, int W, int Ho, int Wo, hipStream_t st) {
  const int CPR = p.C / 8, RPS = 256 / CPR;
  const int Q = N * ((H + 1) / 2) * ((W + 1) / 2);
  int sweeps = (Q + RPS - 1) / RPS;
  int spw = sweeps / 2048;
  spw = spw < 2 ? 2 : (spw > 8 ? 8 : spw);
  const int gpb = spw * RPS;
  const int grid = (Q + gpb - 1) / gpb;
  DISPATCH(dtype,
           hipLaunchKernelGGL(stem_bn_pool_bwd_apply4_kernel<bf16>, dim3(grid), dim3(256), 0, st, p, (const bf16*)dpool, idx, (const bf16*)y, dstats, (bf16*)dy, dgamma, dbeta, N, H, W, Ho, Wo, gpb),
           hipLaunchKernelGGL(stem_bn_pool_bwd_apply4_kernel<float>, dim3(grid), dim3(256), 0, st, p, (const float*)dpool, idx, (const float*)y, dstats, (float*)dy, dgamma, dbeta, N, H, W, Ho, Wo, gpb));
  return (int)hipGetLastError();
}

extern "C" int clite_stem_bn_pool_fwd_ex(const clite_bn* p, int dtype, const void* y, void* pooled, uint8_t* idx, void* ymax, int N, int H, int W, void* stream) {
  if (!p || !y || !pooled || !idx || N <= 0 || p->M != N * H * W || !bn_ok(p->M, p->C) || p->replicas < 1) return -1;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  int rpb;
  int grid = bn_grid(N * Ho * Wo, p->C, &rpb);
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype,
           hipLaunchKernelGGL(stem_bn_pool_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, *p, (const bf16*)y, (bf16*)pooled, idx, (bf16*)ymax, N, H, W, Ho, Wo, rpb),
           hipLaunchKernelGGL(stem_bn_pool_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, *p, (const float*)y, (float*)pooled, idx, (float*)ymax, N, H, W, Ho, Wo, rpb));
  return (int)hipGetLastError();
}
extern "C" int clite_stem_bn_pool_fwd(const clite_bn* p, int dtype, const void* y, void* pooled, uint8_t* idx, int N, int H, int W, void* stream) {
  if (p && p->relu_bits) return -1;          // (the _ex entry point writes them)
  return clite_stem_bn_pool_fwd_ex(p, dtype, y, pooled, idx, nullptr, N, H, W, stream);
}

extern "C" int clite_stem_bn_pool_bwd(const clite_bn* p, int dtype, const void* dpool, const uint8_t* idx, const void* y, float* dstats, void* dy,
                                      float* dgamma, float* dbeta, int N, int H, int W, void* stream) {
  if (!p || !dpool || !idx || !y || !dstats || !dy || N <= 0 || p->M != N * H * W || !bn_ok(p->M, p->C) || p->replicas < 1) return -1;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  int rpb_r, rpb_a;
  const int grid_r = bn_grid_reduce(p->M, p->C, p->replicas, &rpb_r);
  const int grid_a = bn_grid(p->M, p->C, &rpb_a);
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype,
           hipLaunchKernelGGL(stem_bn_pool_bwd_reduce_kernel<bf16>, dim3(grid_r), dim3(256), 0, st, *p, (const bf16*)dpool, idx, (const bf16*)y, dstats, N, H, W, Ho, Wo, rpb_r),
           hipLaunchKernelGGL(stem_bn_pool_bwd_reduce_kernel<float>, dim3(grid_r), dim3(256), 0, st, *p, (const float*)dpool, idx, (const float*)y, dstats, N, H, W, Ho, Wo, rpb_r));
  (void)grid_a; (void)rpb_a;
  return stem_apply4(*p, dtype, dpool, idx, y, (const float*)dstats, dy, dgamma, dbeta, N, H, W, Ho, Wo, st);
}

extern "C" int clite_stem_bn_pool_bwd_apply(const clite_bn* p, int dtype, const void* dpool, const uint8_t* idx, const void* y, const float* dstats, void* dy,
                                            float* dgamma, float* dbeta, int N, int H, int W, void* stream) {
  if (!p || !dpool || !idx || !y || !dstats || !dy || N <= 0 || p->M != N * H * W || !bn_ok(p->M, p->C) || p->replicas < 1) return -1;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipStream_t st = (hipStream_t)stream;
  return stem_apply4(*p, dtype, dpool, idx, y, dstats, dy, dgamma, dbeta, N, H, W, Ho, Wo, st);
}

extern "C" int clite_avgpool_fwd(int dtype, const void* x, void* out, int N, int HW, int C, void* stream) {
  if (C % 8 || N <= 0) return -1;
  int grid = (N * (C / 8) + 255) / 256;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype,
           hipLaunchKernelGGL(avgpool_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)x, (bf16*)out, N, HW, C),
           hipLaunchKernelGGL(avgpool_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, (float*)out, N, HW, C));
  return (int)hipGetLastError();
}

extern "C" int clite_avgpool_bwd(int dtype, const void* dout, void* dx, int N, int HW, int C, void* stream) {
  if (C % 8 || N <= 0) return -1;
  int grid = ew_grid((size_t)N * HW * (C / 8));
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype,
           hipLaunchKernelGGL(avgpool_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)dout, (bf16*)dx, N, HW, C),
           hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)dout, (float*)dx, N, HW, C));
  return (int)hipGetLastError();
}

extern "C" int clite_image_to_nhwc4(int dtype, const float* img, void* out, int N, int H, int W, int pad, int Hp, int Wp, void* stream) {
  if (N <= 0 || Hp < H + 2 * pad || Wp < W + 2 * pad) return -1;
  if ((size_t)N * Hp >= ((size_t)1 << 31)) return -1;
  int grid = (N * Hp + 3) / 4;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype,
           hipLaunchKernelGGL(image_to_nhwc4_kernel<bf16>, dim3(grid), dim3(256), 0, st, img, (bf16*)out, N, H, W, pad, Hp, Wp),
           hipLaunchKernelGGL(image_to_nhwc4_kernel<float>, dim3(grid), dim3(256), 0, st, img, (float*)out, N, H, W, pad, Hp, Wp));
  return (int)hipGetLastError();
}

extern "C" int clite_colsum(int dtype, const void* x, float* out, int M, int N, void* stream) {
  if (M <= 0 || N <= 0 || N % 8) return -1;
  // A narrow matrix (N = 64 / 128: the folded BatchNorm backward's colsum(a) over 100k - 400k pixels) would use a quarter / half of each workgroup's
  // 256 column lanes: 256 / N consecutive rows are one 256-wide row of the same memory, and column j of that view belongs to column j % N
  int nfold = 0;
  if (N < 256 && 256 % N == 0 && M % (256 / N) == 0 && M / (256 / N) >= 64) { nfold = N; M /= 256 / N; N = 256; }
  int gx = (N / 8 + 31) / 32;
  int slabs = 256 / gx;           // ~256 workgroups; every slab ends in one float atomic per column (a few dozen per address)
  if (slabs < 8) slabs = 8;
  if (slabs > 64) slabs = 64;
  if (nfold && M >= 64 * 1024) slabs = 1024;          // tens of MB: enough workgroups to stream at the HBM rate (4 x 1024 atomics per address)
  if (slabs > M) slabs = M;
  if (clite::deterministic()) slabs = 1;      // det.h: one contribution per column
  int rpb = (M + slabs - 1) / slabs;
  slabs = (M + rpb - 1) / rpb;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype,
           hipLaunchKernelGGL(colsum_kernel<bf16>, dim3(gx, slabs), dim3(256), 0, st, (const bf16*)x, out, M, N, rpb, nfold),
           hipLaunchKernelGGL(colsum_kernel<float>, dim3(gx, slabs), dim3(256), 0, st, (const float*)x, out, M, N, rpb, nfold));
  return (int)hipGetLastError();
}
