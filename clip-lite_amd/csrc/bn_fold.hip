// The block-output BatchNorm backward folded into its consumers (include/clite.h, ABI v12): the two small kernels around the GEMMs.
//
// BatchNorm backward is linear in the masked gradient dz: dy = ka . dz + kb + kc . (y - mean) per channel, with ka, kb, kc known once the two
// reductions S1 = sum dz, S2 = sum dz (y - mean) are (they come out of the epilogue of the GEMM that wrote dz). For the 1 x 1 convolution y = a W^T
// in front of that BatchNorm the apply pass therefore folds into
//   * the input gradient: one GEMM over the K-concatenation [dz | y] against the row-scaled weights [diag(ka) W ; diag(kc) W], + a constant row
//     (clite_conv_dgrad_bnfold in gemm.hip; bn_fold_prepare_kernel makes the weights and the row), and
//   * the weight gradient: diag(ka) (dz^T a) — the grouped launch's member with clite_wgrad_item.row_scale — + kb (x) colsum(a) + diag(kc) W Cov(a)
//     (bn_fold_wgrad_finish_kernel adds both correction terms in f32; colsum(a) comes from the clite_bn_apply that wrote a: clite_bn.out_sum).
// dy — the largest tensors of the ResNet backward, 205 MB per layer1 block at batch 128 — is neither written nor read.
#include "vec.h"
#include "clite.h"

using namespace clite;

namespace {

// one workgroup per input channel c of the convolution (= output column of its input gradient): row c of w2, bias[c]
__global__ __launch_bounds__(256) void bn_fold_prepare_kernel(clite_bn p, const float* __restrict__ dstats, const bf16* __restrict__ wt, int Cin, bf16* __restrict__ w2,
                                                              float* __restrict__ bias, float* __restrict__ coef, float* dgamma, float* dbeta) {
  __shared__ float red[4];
  const int c = blockIdx.x, K = p.C, tid = threadIdx.x;
  const float inv_count = 1.0f / (float)p.M;
  float acc = 0.f;
  for (int k = tid; k < K; k += 256) {
    float v1 = 0.f, v2 = 0.f, S1 = 0.f, S2 = 0.f;
    for (int r = 0; r < p.replicas; ++r) {
      const float* st = p.stats + (size_t)r * p.rstride;
      const float* ds = dstats + (size_t)r * p.rstride;
      v1 += st[k];
      v2 += st[(p.centered ? 2 : 1) * K + k];
      S1 += ds[k];
      S2 += ds[K + k];
    }
    // (the same arithmetic, in the same order, as bn_bwd_apply_kernel's prologue)
    const float mean = v1 * inv_count;
    const float var = p.centered ? v2 * inv_count : fmaxf(v2 * inv_count - mean * mean, 0.f);
    const float rstd = rsqrtf(var + p.eps);
    const float G = rstd * S2;
    const float a = p.gamma[k] * rstd;
    const float ka = a, kc = -a * rstd * G * inv_count, kb = -a * S1 * inv_count;
    const float w = bf2f(wt[(size_t)c * K + k]);
    const bf16 wy = f2bf(kc * w), wz = f2bf(ka * w);
    w2[((size_t)c * 2 + 0) * K + k] = wy;          // r = 0 of the dgrad gather reads slot 1 of the pair buffer: y
    w2[((size_t)c * 2 + 1) * K + k] = wz;          // r = 1 reads slot 0: dz
    // the constant row against the ROUNDED weights of the y term: sum_k (y - mean) w2 cancels then exactly where y - mean does
    acc += kb * w - mean * bf2f(wy);
    if (c == 0) {
      coef[k] = ka; coef[K + k] = kb; coef[2 * K + k] = kc;
      if (dgamma) dgamma[k] += G;
      if (dbeta) dbeta[k] += S1;
    }
  }
#pragma unroll
  for (int sh = 32; sh >= 1; sh >>= 1) acc += wave_shfl_xor(acc, sh);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) bias[c] = red[0] + red[1] + red[2] + red[3];
}

// dw[k][c] += kb[k] s[c] + kc[k] (sum_c' W[k][c'] G[c'][c] - (sum_c' W[k][c'] s[c']) s[c] / M): one thread per (k, c), c fastest — G's row c' is read
// coalesced, W[k][c'] = wt[c'][k] is a broadcast among the threads that share k; both matrices are a few hundred KB (L2-resident after the first touch),
// the loop is unrolled so that eight independent pairs of loads are in flight (a loop of dependent round trips made this kernel 76 - 140 us in the step)
__global__ __launch_bounds__(256) void bn_fold_wgrad_finish_kernel(const float* __restrict__ G, const float* __restrict__ asum, int R, int rstride, const float* __restrict__ coef,
                                                                   const bf16* __restrict__ wt, float inv_count, int K, int Cin, float* __restrict__ dw) {
  __shared__ float sall[2048];
  const int tid = threadIdx.x;
  for (int c = tid; c < Cin; c += 256) sall[c] = 0.f;
  __syncthreads();
#pragma unroll 8
  for (int i = tid; i < R * Cin; i += 256) {          // colsum(a): the replicas folded once per workgroup
    const int r = i / Cin, c = i - r * Cin;
    atomicAdd(&sall[c], asum[(size_t)r * rstride + c]);
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + tid;
  if (i >= K * Cin) return;
  const int k = i / Cin, c = i - k * Cin;
  float wg = 0.f, ws = 0.f;
#pragma unroll 8
  for (int cp = 0; cp < Cin; ++cp) {
    const float w = bf2f(wt[(size_t)cp * K + k]);
    wg += w * G[(size_t)cp * Cin + c];
    ws += w * sall[cp];
  }
  const float sc = sall[c];
  atomic_add_f32(dw + i, coef[K + k] * sc + coef[2 * K + k] * (wg - ws * sc * inv_count));
}

}  // namespace

extern "C" int clite_bn_fold_prepare(const clite_bn* p, const float* dstats, const void* wt, int Cin, void* w2, float* bias, float* coef, float* dgamma, float* dbeta,
                                     void* stream) {
  if (!p || !dstats || !wt || !w2 || !bias || !coef || Cin <= 0 || Cin % 8 || p->C <= 0 || p->C % 8 || p->M <= 0 || p->replicas < 1 || !p->stats || !p->gamma)
    return -1;
  hipLaunchKernelGGL(bn_fold_prepare_kernel, dim3(Cin), dim3(256), 0, (hipStream_t)stream, *p, dstats, (const bf16*)wt, Cin, (bf16*)w2, bias, coef, dgamma, dbeta);
  return (int)hipGetLastError();
}

extern "C" int clite_bn_fold_wgrad_finish(const float* G, const float* asum, int asum_replicas, int asum_stride, const float* coef, const void* wt, int M, int K, int Cin,
                                          float* dw, void* stream) {
  if (!G || !asum || asum_replicas < 1 || !coef || !wt || !dw || M <= 0 || K <= 0 || Cin <= 0 || Cin > 2048 || (size_t)K * Cin >= ((size_t)1 << 31)) return -1;
  hipLaunchKernelGGL(bn_fold_wgrad_finish_kernel, dim3((K * Cin + 255) / 256), dim3(256), 0, (hipStream_t)stream, G, asum, asum_replicas, asum_stride, coef, (const bf16*)wt,
                     1.0f / (float)M, K, Cin, dw);
  return (int)hipGetLastError();
}
