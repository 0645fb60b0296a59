// 8-element vector load/store helpers shared by all kernels: activations are stored as bf16 (production) or f32
// (exact parity mode); all arithmetic happens in f32 registers.
#ifndef CLITE_VEC_H
#define CLITE_VEC_H
#include "intrin.h"

namespace clite {

// 8 consecutive elements <-> fp32 registers
DEV void load8(const bf16* p, float (&v)[8]) {
  Chunk16 c;
  c.u = *(const u32x4*)p;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = bf2f(c.e[e]);
}
DEV void load8(const float* p, float (&v)[8]) {
  f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
// streaming forms: the tensor is not read again soon (last use of an activation in backward) / not re-read by this kernel
DEV void load8_nt(const bf16* p, float (&v)[8]) {
  Chunk16 c;
  c.u = __builtin_nontemporal_load((const u32x4*)p);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = bf2f(c.e[e]);
}
DEV void load8_nt(const float* p, float (&v)[8]) {
  f32x4 a = __builtin_nontemporal_load((const f32x4*)p), b = __builtin_nontemporal_load((const f32x4*)(p + 4));
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
DEV void store8_nt(bf16* p, const float (&v)[8]) {
  Chunk16 c;
#pragma unroll
  for (int e = 0; e < 8; ++e) c.e[e] = f2bf(v[e]);
  __builtin_nontemporal_store(c.u, (u32x4*)p);
}
DEV void store8_nt(float* p, const float (&v)[8]) {
  __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, (f32x4*)p);
  __builtin_nontemporal_store(f32x4{v[4], v[5], v[6], v[7]}, (f32x4*)(p + 4));
}
DEV void store8(bf16* p, const float (&v)[8]) {
  Chunk16 c;
#pragma unroll
  for (int e = 0; e < 8; ++e) c.e[e] = f2bf(v[e]);
  *(u32x4*)p = c.u;
}
DEV void store8(float* p, const float (&v)[8]) {
  *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]};
  *(f32x4*)(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
DEV void round8_bf16(float (&v)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = bf2f(f2bf(v[e]));
}

// 8 consecutive elements held raw (as loaded) until they are needed: lets a kernel issue several loads before converting any
template <typename T> struct Raw8;
// size of a buffer resource over an operand whose extent the epilogue does not know (every tensor of the step is < 0xF0000000 bytes: gemm.hip
// fits32); an absent operand is addressed with OOB_OFF, which reads zeros from any resource
constexpr uint32_t RSRC_WHOLE = 0xF0000000u;

template <> struct Raw8<bf16> {
  u32x4 a;
  DEV void ld(const bf16* p) { a = *(const u32x4*)p; }
  DEV void ldb(rsrc_t r, uint32_t byte_off) { a = buf_load16(r, byte_off); }          // branch-free form: out-of-range offset -> zeros
  DEV void get(float (&v)[8]) const {
    Chunk16 c;
    c.u = a;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = bf2f(c.e[e]);
  }
};
template <> struct Raw8<float> {
  f32x4 a, b;
  DEV void ld(const float* p) { a = *(const f32x4*)p; b = *(const f32x4*)(p + 4); }
  DEV void ldb(rsrc_t r, uint32_t byte_off) {
    union { u32x4 u; f32x4 f; } x, y;
    x.u = buf_load16(r, byte_off);
    y.u = buf_load16(r, byte_off == OOB_OFF ? OOB_OFF : byte_off + 16);
    a = x.f; b = y.f;
  }
  DEV void get(float (&v)[8]) const {
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
  }
};

// ReLU that keeps a NaN a NaN, like torch.relu (fmaxf(x, 0) returns 0 for a NaN: a diverged activation would silently become a zero and the
// loss stay finite — found by planting a NaN in an input image, tests/test_gpu_fp8.py)
DEV float relu_f(float x) { return x < 0.f ? 0.f : x; }

DEV void zero8(float (&v)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
}

template <typename T> struct DType;
template <> struct DType<bf16> { static constexpr int id = 0; };
template <> struct DType<float> { static constexpr int id = 1; };

}  // namespace clite
#endif
