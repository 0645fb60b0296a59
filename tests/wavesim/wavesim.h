// wavesim — TEST INFRASTRUCTURE ONLY. Never linked into the shipped library.
//
// A lane-accurate host emulation of the small HIP/gfx950 subset that the kernels in
// clip-lite_amd/csrc use, so kernel index math (LDS images, MFMA operand maps,
// transposed LDS reads, buffer bounds) can be debugged on the CPU build box, which
// has no GPU, and run under host AddressSanitizer. One OS thread per GPU thread,
// workgroups run one after another; wave-collective operations (MFMA, ds_read_tr,
// shuffles) rendezvous on a per-wave barrier.
//
// Built by tests/wavesim/Makefile with `clang++ -x c++ -include wavesim.h`; this
// header pre-defines CLITE_INTRIN_H so csrc/intrin.h's device definitions are
// skipped and the emulations below are used instead.
#ifndef CLITE_WAVESIM_H
#define CLITE_WAVESIM_H
#define CLITE_INTRIN_H  // shadow csrc/intrin.h

#include <atomic>
#include <barrier>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__
#define DEV inline __attribute__((always_inline))
#define WAVE 64

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
  memset(p, v, n);
  return hipSuccess;
}
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
inline hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t) {      // host memory both sides here
  memcpy(dst, src, n);
  return hipSuccess;
}

namespace wavesim {
struct WaveCtx {
  std::unique_ptr<std::barrier<>> bar;
  uint64_t slot[64][8];  // per-lane exchange area (up to 64 B per lane)
};
struct BlockCtx {
  std::unique_ptr<std::barrier<>> bar;
  std::vector<WaveCtx> waves;
};
extern thread_local BlockCtx* g_block;
extern thread_local int g_lane;
extern thread_local int g_wave;
}  // namespace wavesim

extern thread_local dim3 threadIdx;
extern thread_local dim3 blockIdx;
extern thread_local dim3 blockDim;
extern thread_local dim3 gridDim;

inline void __syncthreads() { wavesim::g_block->bar->arrive_and_wait(); }
inline void wave_barrier_() { wavesim::g_block->waves[wavesim::g_wave].bar->arrive_and_wait(); }

template <class K, class... Args>
void wavesim_launch(K kernel, dim3 grid, dim3 block, Args... args) {
  unsigned nthreads = block.x * block.y * block.z;
  assert(block.y == 1 && block.z == 1 && "wavesim: 1-D blocks only");
  assert(nthreads % 64 == 0 && "wavesim: block size must be a multiple of 64");
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        wavesim::BlockCtx ctx;
        ctx.bar = std::make_unique<std::barrier<>>(nthreads);
        ctx.waves.resize(nthreads / 64);
        for (auto& w : ctx.waves) w.bar = std::make_unique<std::barrier<>>(64);
        std::vector<std::thread> ts;
        ts.reserve(nthreads);
        for (unsigned t = 0; t < nthreads; ++t) {
          ts.emplace_back([&, t]() {
            threadIdx = dim3(t, 0, 0);
            blockIdx = dim3(bx, by, bz);
            blockDim = block;
            gridDim = grid;
            wavesim::g_block = &ctx;
            wavesim::g_lane = t & 63;
            wavesim::g_wave = t >> 6;
            kernel(args...);
          });
        }
        for (auto& th : ts) th.join();
      }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  wavesim_launch(kernel, dim3(grid), dim3(block), ##__VA_ARGS__)

// ---------------------------------------------------------------- types
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

union Chunk16 {
  u32x4 u;
  bf16x8 h;
  bf16 e[8];
};

// ---------------------------------------------------------------- buffer resources
struct rsrc_t {
  char* base;
  uint32_t bytes;
};
#define OOB_OFF 0xFFFFFFF0u
inline rsrc_t make_rsrc(const void* p, uint32_t bytes) { return rsrc_t{(char*)p, bytes}; }
inline bool wavesim_in_range(rsrc_t r, uint32_t off, uint32_t n, const char* what) {
  if ((uint64_t)off + n <= r.bytes) return true;
  // Kernels mark intentional zero-fill with OOB_OFF-based offsets (>= 0x80000000).
  // Any other out-of-range access is an indexing bug: fail loudly in the simulator.
  if (off < 0x80000000u) {
    fprintf(stderr, "wavesim: unintended out-of-range buffer %s: off=%u n=%u size=%u (block %u thread %u)\n",
            what, off, n, r.bytes, blockIdx.x, threadIdx.x);
    abort();
  }
  return false;
}
template <class T>
inline T wavesim_buf_load(rsrc_t r, uint32_t off) {
  T v;
  memset(&v, 0, sizeof(T));
  if (wavesim_in_range(r, off, sizeof(T), "load")) memcpy(&v, r.base + off, sizeof(T));
  return v;
}
template <class T>
inline void wavesim_buf_store(rsrc_t r, uint32_t off, T v) {
  if (wavesim_in_range(r, off, sizeof(T), "store")) memcpy(r.base + off, &v, sizeof(T));
}
inline u32x4 buf_load16(rsrc_t r, uint32_t off) { assert(off % 4 == 0); return wavesim_buf_load<u32x4>(r, off); }
inline u32x2 buf_load8(rsrc_t r, uint32_t off) { assert(off % 4 == 0); return wavesim_buf_load<u32x2>(r, off); }
inline uint32_t buf_load4(rsrc_t r, uint32_t off) { assert(off % 4 == 0); return wavesim_buf_load<uint32_t>(r, off); }
inline uint32_t buf_load1(rsrc_t r, uint32_t off) { return (uint32_t)wavesim_buf_load<uint8_t>(r, off); }
inline void buf_store16(rsrc_t r, uint32_t off, u32x4 v) { assert(off % 4 == 0); wavesim_buf_store(r, off, v); }
inline void buf_store8(rsrc_t r, uint32_t off, u32x2 v) { assert(off % 4 == 0); wavesim_buf_store(r, off, v); }
inline void buf_store4(rsrc_t r, uint32_t off, uint32_t v) { assert(off % 4 == 0); wavesim_buf_store(r, off, v); }

// LDS-DMA is emulated synchronously (data lands immediately), so the simulator checks addressing and swizzles, not the
// vmcnt/barrier protocol.
inline void buf_load16_lds(rsrc_t r, uint32_t off, void* lds_wave_base) {
  u32x4 v = buf_load16(r, off);
  memcpy((char*)lds_wave_base + wavesim::g_lane * 16, &v, 16);
}
inline void buf_load16_lds_nt(rsrc_t r, uint32_t off, void* lds_wave_base) { buf_load16_lds(r, off, lds_wave_base); }
template <int N> inline void wait_vmcnt() {}
inline void opaque_i(int&) {}
inline void barrier_raw() { __syncthreads(); }
inline void lds_barrier() { __syncthreads(); }

// ---------------------------------------------------------------- conversions
inline float bf2f(bf16 x) {
  uint16_t b;
  memcpy(&b, &x, 2);
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
inline bf16 f2bf(float f) {  // round-to-nearest-even, NaN preserved
  uint32_t u;
  memcpy(&u, &f, 4);
  uint16_t b;
  if ((u & 0x7fffffffu) > 0x7f800000u) b = (uint16_t)((u >> 16) | 0x40);
  else b = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  bf16 r;
  memcpy(&r, &b, 2);
  return r;
}

// value known to be identical in every lane of the wave (the device build moves it to a scalar register)
inline int wave_uniform(int v) { return v; }

// ---------------------------------------------------------------- wave collectives
inline float wave_shfl(float v, int src) {
  auto& w = wavesim::g_block->waves[wavesim::g_wave];
  memcpy(&w.slot[wavesim::g_lane][0], &v, 4);
  wave_barrier_();
  float r;
  memcpy(&r, &w.slot[src & 63][0], 4);
  wave_barrier_();
  return r;
}
inline float wave_shfl_xor(float v, int m) { return wave_shfl(v, wavesim::g_lane ^ m); }
inline int wave_shfl_xor_i(int v, int m) {
  float f;
  memcpy(&f, &v, 4);
  f = wave_shfl_xor(f, m);
  int r;
  memcpy(&r, &f, 4);
  return r;
}
inline float wave_sum(float v) {
  for (int m = 32; m >= 1; m >>= 1) v += wave_shfl_xor(v, m);
  return v;
}
inline float wave_max(float v) {
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, wave_shfl_xor(v, m));
  return v;
}

// v_mfma_f32_32x32x16_bf16, operand maps per cdna_hip_programming.md §3.
inline f32x16 mfma32_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  auto& w = wavesim::g_block->waves[wavesim::g_wave];
  int l = wavesim::g_lane;
  memcpy(&w.slot[l][0], &a, 16);
  memcpy(&w.slot[l][2], &b, 16);
  wave_barrier_();
  int col = l & 31;
  for (int i = 0; i < 16; ++i) {
    int row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
    float acc = c[i];
    for (int k = 0; k < 16; ++k) {
      bf16x8 av, bv;
      memcpy(&av, &w.slot[row + 32 * (k >> 3)][0], 16);  // A[row][k] lives in lane row+32*(k/8), elem k%8
      memcpy(&bv, &w.slot[col + 32 * (k >> 3)][2], 16);  // B[k][col] lives in lane col+32*(k/8), elem k%8
      acc = fmaf(bf2f(av[k & 7]), bf2f(bv[k & 7]), acc);
    }
    c[i] = acc;
  }
  wave_barrier_();
  return c;
}
// OCP e4m3fn: 1 sign, 4 exponent (bias 7), 3 mantissa bits; no infinities, 0x7f / 0xff = NaN, max 448
inline float fp8_e4m3_to_f32(uint8_t v) {
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float r;
  if (e == 15 && m == 7) r = NAN;
  else if (e == 0) r = ldexpf((float)m, -9);               // subnormal: m/8 * 2^-6
  else r = ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -r : r;
}
inline uint8_t f32_to_fp8_e4m3(float f) {                  // round to nearest even, saturating at +-448 (callers clamp anyway)
  if (f != f) return 0x7f;
  uint8_t s = std::signbit(f) ? 0x80 : 0;
  float a = fabsf(f);
  if (a >= 464.f) return s | 0x7e;                         // beyond the midpoint to the next (non-existent) value: saturate
  if (a < ldexpf(1.f, -10)) return s;                      // below half the smallest subnormal
  int e;
  float m = frexpf(a, &e);                                 // a = m * 2^e, m in [0.5, 1)
  int E = e - 1 + 7;                                       // biased exponent if normal
  float q;
  if (E >= 1) q = rintf((m * 2.f - 1.f) * 8.f);            // mantissa steps of 1/8 above 1.0
  else { q = rintf(ldexpf(a, 9)); E = 0; }                 // subnormal: multiples of 2^-9
  int mi = (int)q;
  if (E >= 1 && mi == 8) { mi = 0; ++E; }
  if (E == 0 && mi == 8) { mi = 0; E = 1; }
  if (E > 15 || (E == 15 && mi == 7)) return s | 0x7e;
  return s | (uint8_t)(E << 3) | (uint8_t)mi;
}
inline uint32_t cvt2_fp8(float a, float b) { return (uint32_t)f32_to_fp8_e4m3(a) | ((uint32_t)f32_to_fp8_e4m3(b) << 8); }
// v_mfma_f32_32x32x16_fp8_fp8: operand maps as the bf16 form, one byte per element
inline f32x16 mfma32_fp8(uint64_t a, uint64_t b, f32x16 c) {
  auto& w = wavesim::g_block->waves[wavesim::g_wave];
  int l = wavesim::g_lane;
  w.slot[l][0] = a;
  w.slot[l][1] = b;
  wave_barrier_();
  int col = l & 31;
  for (int i = 0; i < 16; ++i) {
    int row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
    float acc = c[i];
    for (int k = 0; k < 16; ++k) {
      uint8_t av = (uint8_t)(w.slot[row + 32 * (k >> 3)][0] >> (8 * (k & 7)));
      uint8_t bv = (uint8_t)(w.slot[col + 32 * (k >> 3)][1] >> (8 * (k & 7)));
      acc = fmaf(fp8_e4m3_to_f32(av), fp8_e4m3_to_f32(bv), acc);
    }
    c[i] = acc;
  }
  wave_barrier_();
  return c;
}
// OCP e5m2: 1 sign, 5 exponent (bias 15), 2 mantissa bits; infinities and NaNs as IEEE half's top byte
inline float fp8_e5m2_to_f32(uint8_t v) {
  int s = v >> 7, e = (v >> 2) & 31, m = v & 3;
  float r;
  if (e == 31) r = m ? NAN : INFINITY;
  else if (e == 0) r = ldexpf((float)m, -16);              // subnormal: m/4 * 2^-14
  else r = ldexpf(1.0f + m / 4.0f, e - 15);
  return s ? -r : r;
}
inline uint8_t f32_to_fp8_e5m2(float f) {                  // round to nearest even; overflow saturates at +-57344 (v_cvt_pk_bf8_f32 with clamping semantics of the callers' range)
  if (f != f) return 0x7f;
  uint8_t s = std::signbit(f) ? 0x80 : 0;
  float a = fabsf(f);
  if (a >= 61440.f) return s | 0x7b;                       // beyond the midpoint above the largest finite value (57344): saturate
  if (a < ldexpf(1.f, -17)) return s;
  int e;
  float m = frexpf(a, &e);
  int E = e - 1 + 15;
  float q;
  if (E >= 1) q = rintf((m * 2.f - 1.f) * 4.f);
  else { q = rintf(ldexpf(a, 16)); E = 0; }
  int mi = (int)q;
  if (E >= 1 && mi == 4) { mi = 0; ++E; }
  if (E == 0 && mi == 4) { mi = 0; E = 1; }
  if (E >= 31) return s | 0x7b;
  return s | (uint8_t)(E << 2) | (uint8_t)mi;
}
inline uint32_t cvt2_bf8(float a, float b) { return (uint32_t)f32_to_fp8_e5m2(a) | ((uint32_t)f32_to_fp8_e5m2(b) << 8); }
// v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 x e4m3, unit scales): lane (r, h) holds k = 16 h + j (bytes 0..15) and k = 32 + 16 h + j - 16 (bytes 16..31)
// - the map tools/micro/mfma_scale_probe.hip measured on the hardware
template <bool A_E5M2>
inline f32x16 mfma32x64_f8_(u32x4 a_lo, u32x4 a_hi, u32x4 b_lo, u32x4 b_hi, f32x16 c) {
  auto& w = wavesim::g_block->waves[wavesim::g_wave];
  int l = wavesim::g_lane;
  memcpy(&w.slot[l][0], &a_lo, 16);
  memcpy(&w.slot[l][2], &a_hi, 16);
  memcpy(&w.slot[l][4], &b_lo, 16);
  memcpy(&w.slot[l][6], &b_hi, 16);
  wave_barrier_();
  int col = l & 31;
  for (int i = 0; i < 16; ++i) {
    int row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
    float acc = c[i];
    for (int k = 0; k < 64; ++k) {
      const int h = (k >> 4) & 1, j = (k & 15) + 16 * (k >> 5);
      const uint8_t av = ((const uint8_t*)&w.slot[row + 32 * h][0])[j];
      const uint8_t bv = ((const uint8_t*)&w.slot[col + 32 * h][4])[j];
      acc = fmaf(A_E5M2 ? fp8_e5m2_to_f32(av) : fp8_e4m3_to_f32(av), fp8_e4m3_to_f32(bv), acc);
    }
    c[i] = acc;
  }
  wave_barrier_();
  return c;
}
inline f32x16 mfma32x64_fp8(u32x4 a_lo, u32x4 a_hi, u32x4 b_lo, u32x4 b_hi, f32x16 c) { return mfma32x64_f8_<false>(a_lo, a_hi, b_lo, b_hi, c); }
inline f32x16 mfma32x64_bf8_fp8(u32x4 a_lo, u32x4 a_hi, u32x4 b_lo, u32x4 b_hi, f32x16 c) { return mfma32x64_f8_<true>(a_lo, a_hi, b_lo, b_hi, c); }
// v_mfma_f32_32x32x2_f32
inline f32x16 mfma32_f32(float a, float b, f32x16 c) {
  auto& w = wavesim::g_block->waves[wavesim::g_wave];
  int l = wavesim::g_lane;
  memcpy(&w.slot[l][0], &a, 4);
  memcpy(&w.slot[l][1], &b, 4);
  wave_barrier_();
  int col = l & 31;
  for (int i = 0; i < 16; ++i) {
    int row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
    float acc = c[i];
    for (int k = 0; k < 2; ++k) {
      float av, bv;
      memcpy(&av, &w.slot[row + 32 * k][0], 4);
      memcpy(&bv, &w.slot[col + 32 * k][1], 4);
      acc = fmaf(av, bv, acc);
    }
    c[i] = acc;
  }
  wave_barrier_();
  return c;
}

// ds_read_b64_tr_b16 per cdna_hip_programming.md §5.5 T10: inside each group of 16
// consecutive lanes, lane 4q+p supplies the address of row q, columns 4p..4p+3 of a
// 4x16 block of 16-bit elements; lane i receives column i, row q in element q.
inline s16x4 lds_read_tr16(const void* p) {
  auto& w = wavesim::g_block->waves[wavesim::g_wave];
  int l = wavesim::g_lane;
  assert(((uintptr_t)p & 7) == 0 && "ds_read_b64_tr_b16 needs an 8-byte aligned address");
  uint64_t addr = (uint64_t)(uintptr_t)p;
  w.slot[l][0] = addr;
  wave_barrier_();
  int g = l & ~15, i = l & 15;
  s16x4 r;
  for (int q = 0; q < 4; ++q) {
    int src_lane = g + 4 * q + (i >> 2);
    const short* row = (const short*)(uintptr_t)w.slot[src_lane][0];
    r[q] = row[i & 3];
  }
  wave_barrier_();
  return r;
}

// ds_read_b64_tr_b8 as measured by tools/micro/tr8_probe.hip: inside each group of 16 lanes, lane s supplies the address of an 8-byte chunk; lane i
// receives in byte j byte (i & 7) of the chunk supplied by lane 2 j + (i >> 3).
inline u32x2 lds_read_tr8(const void* p) {
  auto& w = wavesim::g_block->waves[wavesim::g_wave];
  int l = wavesim::g_lane;
  assert(((uintptr_t)p & 7) == 0 && "ds_read_b64_tr_b8 needs an 8-byte aligned address");
  w.slot[l][0] = (uint64_t)(uintptr_t)p;
  wave_barrier_();
  int g = l & ~15, i = l & 15;
  uint8_t b[8];
  for (int j = 0; j < 8; ++j) {
    const uint8_t* chunk = (const uint8_t*)(uintptr_t)w.slot[g + 2 * j + (i >> 3)][0];
    b[j] = chunk[i & 7];
  }
  wave_barrier_();
  u32x2 r;
  memcpy(&r, b, 8);
  return r;
}

inline void atomic_max_u32(uint32_t* p, uint32_t v) {
  std::atomic_ref<uint32_t> a(*p);
  uint32_t old = a.load();
  while (old < v && !a.compare_exchange_weak(old, v)) {
  }
}
inline uint32_t f32_bits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
inline float bits_f32(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}
inline void atomic_add_f32(float* p, float v) {
  std::atomic_ref<float> a(*p);
  float old = a.load();
  while (!a.compare_exchange_weak(old, old + v)) {
  }
}
inline float atomicAdd(float* p, float v) {
  std::atomic_ref<float> a(*p);
  float old = a.load();
  while (!a.compare_exchange_weak(old, old + v)) {
  }
  return old;
}
inline uint32_t umulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __expf(float x) { return expf(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
inline float __logf(float x) { return logf(x); }

#endif  // CLITE_WAVESIM_H
