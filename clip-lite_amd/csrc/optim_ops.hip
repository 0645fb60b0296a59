// Parameter-update path of the train step (reference train.py:221-226): global-norm gradient clipping
// (torch.nn.utils.clip_grad_norm_), torch.optim.SGD with momentum + weight decay and one (lr, wd) pair per
// parameter tensor (factories.py:464-482), the Lookahead slow/fast interpolation every k-th step
// (optim/lookahead.py:88-101), and the bf16 weight copy the next forward reads — one pass over flat buffers.
// Pure HBM streaming: per element 4 f32 reads (p, g, v, slow on sync steps) and up to 4 writes.
#include "vec.h"
#include "clite.h"

using namespace clite;

namespace {

// Squared gradient norm in two fixed-order stages: every workgroup stores ONE partial (plain store, its own slot), then a single workgroup
// folds the partials in slot order. No float atomics: the result is a pure function of the data, so every data-parallel rank derives the
// same clip factor from the same all-reduced gradients and the replicas stay bit-identical (torch's clip_grad_norm_ has that property;
// an atomically accumulated norm differs in the last bit from rank to rank, and BatchNorm amplifies such differences step after step).
constexpr int SUMSQ_MAX_GRID = 1024;
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* x, size_t n, float* partials) {
  __shared__ float red[4];
  float s = 0.f;
  size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 v = __builtin_nontemporal_load((const f32x4*)(x + i * 4));
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (size_t i = n4 * 4; i < n; ++i) s += x[i] * x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* partials, int n, float* out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] += (red[0] + red[1]) + (red[2] + red[3]);      // `out` is pre-zeroed by contract; += keeps the accumulate semantics
}

// hp: [0] lr multiplier (schedule), [1] momentum, [2] max grad norm (<= 0: no clipping), [3] lookahead sync flag,
//     [4] lookahead alpha, [5] gradient pre-scale (1/world_size after a SUM all-reduce, 1/loss_scale, ...)
template <typename T, bool CAST>
__global__ __launch_bounds__(256) void sgd_step_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ v, float* __restrict__ slow,
                                                       T* __restrict__ cast, const clite_optim_item* __restrict__ items,
                                                       const float* __restrict__ hp, const float* __restrict__ sumsq) {
  const clite_optim_item it = items[blockIdx.x];
  const float lr = it.lr * hp[0], wd = it.wd, mu = hp[1], max_norm = hp[2], alpha = hp[4], gs = hp[5];
  const bool sync = hp[3] != 0.f;
  float clip = 1.f;
  if (max_norm > 0.f) {
    float total = sqrtf(sumsq[0]) * gs;
    clip = fminf(max_norm / (total + 1e-6f), 1.f);
  }
  const float gmul = gs * clip;
  // the five flat arrays never overlap (__restrict__), so the unrolled iterations' loads are all issued before the first store
#pragma unroll 4
  for (uint32_t i = threadIdx.x * 4; i < it.count; i += 1024) {
    size_t o = (size_t)it.start + i;
    // (nontemporal loads / stores measured here: 698 -> 734 us inside the step; plain accesses kept)
    f32x4 pv = *(const f32x4*)(p + o), gv = *(const f32x4*)(g + o), vv = *(const f32x4*)(v + o);
    f32x4 sv = {0.f, 0.f, 0.f, 0.f};
    if (sync) sv = *(const f32x4*)(slow + o);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float ge = gv[e] * gmul + wd * pv[e];
      vv[e] = mu * vv[e] + ge;
      pv[e] = pv[e] - lr * vv[e];
      if (sync) { pv[e] = alpha * pv[e] + (1.f - alpha) * sv[e]; sv[e] = pv[e]; }
    }
    *(f32x4*)(p + o) = pv;
    *(f32x4*)(v + o) = vv;
    *(f32x4*)(g + o) = f32x4{0.f, 0.f, 0.f, 0.f};      // zero_grad for the next step
    if (sync) *(f32x4*)(slow + o) = sv;
    if (CAST) {
      union { bf16 e[4]; u32x2 u; } pk;
#pragma unroll
      for (int e = 0; e < 4; ++e) pk.e[e] = f2bf(pv[e]);
      *(u32x2*)((bf16*)cast + o) = pk.u;
    }
  }
}

// torch.optim.AdamW (reference factories.py:439: OPTIMIZER_NAME "adamw", torch defaults) in the same one-pass form: decoupled weight decay, bias-corrected
// moments, then the Lookahead synchronisation, the gradient zeroing and the bf16 copy as sgd_step_kernel. hp as there, plus [1] beta1, [6] beta2, [7] eps,
// [8] 1 - beta1^t, [9] 1 - beta2^t, [10] 1 - beta1, [11] 1 - beta2 (the host uploads the step's bias corrections with the rest of hp).
template <typename T, bool CAST>
__global__ __launch_bounds__(256) void adamw_step_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v2, float* __restrict__ slow,
                                                         T* __restrict__ cast, const clite_optim_item* __restrict__ items,
                                                         const float* __restrict__ hp, const float* __restrict__ sumsq) {
  const clite_optim_item it = items[blockIdx.x];
  const float lr = it.lr * hp[0], wd = it.wd, max_norm = hp[2], alpha = hp[4], gs = hp[5], b2 = hp[6], eps = hp[7], bc1 = hp[8], bc2 = hp[9], omb1 = hp[10], omb2 = hp[11];
  const bool sync = hp[3] != 0.f;
  float clip = 1.f;
  if (max_norm > 0.f) {
    float total = sqrtf(sumsq[0]) * gs;
    clip = fminf(max_norm / (total + 1e-6f), 1.f);
  }
  const float gmul = gs * clip, decay = 1.f - lr * wd, step_size = lr / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2);
#pragma unroll 4
  for (uint32_t i = threadIdx.x * 4; i < it.count; i += 1024) {
    size_t o = (size_t)it.start + i;
    f32x4 pv = *(const f32x4*)(p + o), gv = *(const f32x4*)(g + o), mv = *(const f32x4*)(m + o), vv = *(const f32x4*)(v2 + o);
    f32x4 sv = {0.f, 0.f, 0.f, 0.f};
    if (sync) sv = *(const f32x4*)(slow + o);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ge = gv[e] * gmul;
      pv[e] *= decay;
      mv[e] = mv[e] + omb1 * (ge - mv[e]);          // exp_avg.lerp_(grad, 1 - beta1)
      vv[e] = b2 * vv[e] + omb2 * ge * ge;          // (1 - beta formed in double on the host, as torch does: 1.f - 0.999f is off by 1.3e-5)
      pv[e] -= step_size * (mv[e] / (sqrtf(vv[e]) * inv_sqrt_bc2 + eps));
      if (sync) { pv[e] = alpha * pv[e] + (1.f - alpha) * sv[e]; sv[e] = pv[e]; }
    }
    *(f32x4*)(p + o) = pv;
    *(f32x4*)(m + o) = mv;
    *(f32x4*)(v2 + o) = vv;
    *(f32x4*)(g + o) = f32x4{0.f, 0.f, 0.f, 0.f};
    if (sync) *(f32x4*)(slow + o) = sv;
    if (CAST) {
      union { bf16 e[4]; u32x2 u; } pk;
#pragma unroll
      for (int e = 0; e < 4; ++e) pk.e[e] = f2bf(pv[e]);
      *(u32x2*)((bf16*)cast + o) = pk.u;
    }
  }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* src, bf16* dst, size_t n) {
  size_t n8 = n / 8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    float v[8];
    load8(src + i * 8, v);
    store8(dst + i * 8, v);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (size_t i = n8 * 8; i < n; ++i) dst[i] = f2bf(src[i]);
}

// Transposed bf16 weight copies for the input-gradient GEMMs. A dgrad contracts over the OUTPUT features, the slow index of a weight stored
// [out][in]: its weight operand would be an XC image (k-rows strided, ds_read_b64_tr_b16 fragments), which the 8-wave kernels stage 10-25 %
// slower than the k-contiguous form (tools/probe_tiles.py: 3840 x 768 x 3072 nn 41 us, the same product as nt 31). The weights change once
// per step, so one grouped launch re-derives [in][out] copies (linear: [N][K] -> [K][N]; conv: [K][R][S][C] -> [C][R][S][K], i.e. R*S
// matrices [K][C] -> [C][K] with row strides R*S*C / R*S*K) and every dgrad runs as a forward-form GEMM. 64 x 64 tiles through LDS
// (odd word pitch: conflict-free both ways), 16-byte global accesses on both sides.
__global__ __launch_bounds__(256) void transpose_weights_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, const clite_transpose_item* __restrict__ items,
                                                                 int n_items) {
  __shared__ uint16_t tile[64][66];
  int lo = 0, hi = n_items - 1;              // item whose [first_tile, first_tile + tiles) holds this workgroup (block-uniform binary search)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].first_tile <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const clite_transpose_item it = items[lo];
  const uint32_t t = blockIdx.x - it.first_tile;
  const uint32_t tiles_c = (it.cols + 63) / 64, tiles_r = (it.rows + 63) / 64;
  const uint32_t b = t / (tiles_r * tiles_c), tr = (t / tiles_c) % tiles_r, tc = t % tiles_c;
  const bf16* s0 = src + it.src_off + (size_t)b * it.src_bstride;
  bf16* d0 = dst + it.dst_off + (size_t)b * it.dst_bstride;
  const int tid = threadIdx.x, c8 = (tid & 7) * 8, r0 = tid >> 3;       // 8 chunks of 8 elements per tile row, 32 rows per sweep
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t r = tr * 64 + r0 + 32 * h, c = tc * 64 + c8;
    Chunk16 ch;
    ch.u = u32x4{0u, 0u, 0u, 0u};
    if (r < it.rows && c < it.cols) ch.u = *(const u32x4*)(s0 + (size_t)r * it.src_ld + c);        // cols % 8 == 0 (checked by the launcher)
    const uint16_t* e = (const uint16_t*)&ch;
#pragma unroll
    for (int k = 0; k < 8; ++k) tile[r0 + 32 * h][c8 + k] = e[k];
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t c = tc * 64 + r0 + 32 * h, r = tr * 64 + c8;          // output row = input column
    if (c < it.cols && r < it.rows) {
      Chunk16 ch;
      uint16_t* e = (uint16_t*)&ch;
#pragma unroll
      for (int k = 0; k < 8; ++k) e[k] = tile[c8 + k][r0 + 32 * h];
      *(u32x4*)(d0 + (size_t)c * it.dst_ld + r) = ch.u;                                             // rows % 8 == 0
    }
  }
}

// dst[i] = src[i] + src[stride + i] + ... + src[(slices - 1) * stride + i], summed in slice order (every rank of a data-parallel job reduces
// ITS chunk of the gradient arena this way, so the order is fixed by construction)
__global__ __launch_bounds__(256) void sum_slices_kernel(const float* __restrict__ src, int slices, size_t stride, size_t n, float* __restrict__ dst) {
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 a = *(const f32x4*)(src + i * 4);
    for (int s = 1; s < slices; ++s) {
      const f32x4 b = *(const f32x4*)(src + (size_t)s * stride + i * 4);
      a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
    }
    *(f32x4*)(dst + i * 4) = a;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = n4 * 4 + threadIdx.x;
    float a = src[i];
    for (int s = 1; s < slices; ++s) a += src[(size_t)s * stride + i];
    dst[i] = a;
  }
}

}  // namespace

extern "C" int clite_sum_slices(const float* src, int slices, uint64_t stride, uint64_t n, float* dst, void* stream) {
  if (!src || !dst || slices < 1 || stride % 4 || ((uintptr_t)src | (uintptr_t)dst) % 16) return -1;
  if (n == 0) return 0;
  size_t g = (n / 4 + 255) / 256;
  int grid = (int)(g < 2048 ? (g ? g : 1) : 2048);
  hipLaunchKernelGGL(sum_slices_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, slices, (size_t)stride, (size_t)n, dst);
  return (int)hipGetLastError();
}

extern "C" int clite_sumsq(const float* x, uint64_t n, float* out, float* partials, int n_partials, void* stream) {
  if (!x || !out || !partials || n_partials < 1) return -1;
  size_t g = (n / 4 + 255) / 256;
  int grid = (int)(g < (size_t)SUMSQ_MAX_GRID ? (g ? g : 1) : SUMSQ_MAX_GRID);
  if (grid > n_partials) grid = n_partials;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, partials);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partials, grid, out);
  return (int)hipGetLastError();
}

extern "C" int clite_sgd_step(float* p, float* g, float* v, float* slow, void* cast_bf16, const clite_optim_item* items, int n_items,
                              const float* hp, const float* sumsq, void* stream) {
  if (!p || !g || !v || !slow || !items || n_items <= 0 || !hp || !sumsq) return -1;
  hipStream_t st = (hipStream_t)stream;
  if (cast_bf16)
    hipLaunchKernelGGL((sgd_step_kernel<bf16, true>), dim3(n_items), dim3(256), 0, st, p, g, v, slow, (bf16*)cast_bf16, items, hp, sumsq);
  else
    hipLaunchKernelGGL((sgd_step_kernel<bf16, false>), dim3(n_items), dim3(256), 0, st, p, g, v, slow, (bf16*)nullptr, items, hp, sumsq);
  return (int)hipGetLastError();
}

extern "C" int clite_adamw_step(float* p, float* g, float* m, float* v2, float* slow, void* cast_bf16, const clite_optim_item* items, int n_items,
                                const float* hp, const float* sumsq, void* stream) {
  if (!p || !g || !m || !v2 || !slow || !items || n_items <= 0 || !hp || !sumsq) return -1;
  hipStream_t st = (hipStream_t)stream;
  if (cast_bf16)
    hipLaunchKernelGGL((adamw_step_kernel<bf16, true>), dim3(n_items), dim3(256), 0, st, p, g, m, v2, slow, (bf16*)cast_bf16, items, hp, sumsq);
  else
    hipLaunchKernelGGL((adamw_step_kernel<bf16, false>), dim3(n_items), dim3(256), 0, st, p, g, m, v2, slow, (bf16*)nullptr, items, hp, sumsq);
  return (int)hipGetLastError();
}

extern "C" int clite_cast_bf16(const float* src, void* dst, uint64_t n, void* stream) {
  if (!src || !dst) return -1;
  size_t g = (n / 8 + 255) / 256;
  int grid = (int)(g < 4096 ? (g ? g : 1) : 4096);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, (size_t)n);
  return (int)hipGetLastError();
}

extern "C" int clite_transpose_weights(const void* src, void* dst, const clite_transpose_item* items_dev, int n_items, uint32_t total_tiles, void* stream) {
  if (!src || !dst || !items_dev || n_items <= 0 || total_tiles == 0) return -1;
  hipLaunchKernelGGL(transpose_weights_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, (bf16*)dst, items_dev, n_items);
  return (int)hipGetLastError();
}

