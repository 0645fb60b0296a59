// Counter-based RNG (Philox4x32-10) for dropout masks and prior noise.
//
// A mask element is a pure function of (seed, site, element index), so the backward pass regenerates
// exactly the forward mask instead of storing it (BERT hidden/attention dropout p=0.1, HF BertModel behind
// reference encoder.py:165-170; prior noise torch.rand_like at reference loss.py:189,196).
// One Philox call yields the 4 uniforms of elements 4q..4q+3: counter = (q_lo, q_hi, site, 0), key = seed.
#ifndef CLITE_RNG_H
#define CLITE_RNG_H
#include "intrin.h"

namespace clite {

struct Philox4 { uint32_t v[4]; };

DEV Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = umulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = umulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  Philox4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

// CLITE_SEED_INDIRECT (include/clite.h): a site with bit 31 set means the seed argument is the device address of a uint64
// holding the seed (wave-uniform scalar load), so a captured hipGraph re-reads the seed of the current step at every replay.
DEV void seed_resolve(uint64_t& seed, uint32_t& site) {
  if (site & 0x80000000u) {
    seed = *(const uint64_t*)(uintptr_t)seed;
    site &= 0x7fffffffu;
  }
}

DEV float u32_to_unit(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }  // [0,1)

// uniforms for elements idx..idx+3 (idx % 4 == 0)
DEV void rng_uniform4(uint64_t seed, uint32_t site, uint64_t idx, float (&u)[4]) {
  uint64_t q = idx >> 2;
  Philox4 p = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), site, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
  for (int e = 0; e < 4; ++e) u[e] = u32_to_unit(p.v[e]);
}
// uniforms for elements idx..idx+7 (idx % 8 == 0), 24 bits each: two Philox calls (prior noise, torch.rand_like)
DEV void rng_uniform8(uint64_t seed, uint32_t site, uint64_t idx, float (&u)[8]) {
  float a[4], b[4];
  rng_uniform4(seed, site, idx, a);
  rng_uniform4(seed, site, idx + 4, b);
#pragma unroll
  for (int e = 0; e < 4; ++e) { u[e] = a[e]; u[4 + e] = b[e]; }
}
// Dropout decisions for elements idx..idx+7 (idx % 8 == 0): 8 uniforms on a 2^-16 grid from ONE Philox call (16 bits per element; counter
// word 3 = 1 keeps the stream apart from rng_uniform4's). `u >= p` then drops with probability ceil(p * 65536) / 65536 (p = 0.1: 0.100006).
// The RNG is the dominant cost of the LayerNorm backward (masks are recomputed, not stored): 27 -> 20 us at M = 3840, C = 768.
DEV void dropout_uniform8(uint64_t seed, uint32_t site, uint64_t idx, float (&u)[8]) {
  uint64_t q = idx >> 3;
  Philox4 p = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), site, 1u, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    u[2 * k] = (float)(p.v[k] & 0xffffu) * (1.0f / 65536.0f);
    u[2 * k + 1] = (float)(p.v[k] >> 16) * (1.0f / 65536.0f);
  }
}
// uniform for a single element (slow path: edges, tiny tensors)
DEV float rng_uniform1(uint64_t seed, uint32_t site, uint64_t idx) {
  float a[4];
  rng_uniform4(seed, site, idx & ~(uint64_t)3, a);
  uint32_t e = (uint32_t)idx & 3u;
  return e == 0 ? a[0] : e == 1 ? a[1] : e == 2 ? a[2] : a[3];
}

}  // namespace clite
#endif
