// The folded BatchNorm backward's input gradient at the 56 x 56 stage (clite_conv_dgrad_bnfold with K = 256, Cin = 64) as a STREAMING kernel.
//
// da [M][64] = [dz | y] [ka W ; kc W] + bias through the BatchNorm-backward epilogue: 1 KB of operand per pixel against 128 B of output — an HBM stream
// with a small matrix product attached. The tile engine runs it at 2.2 TB/s inside the step (profiles/r5_ktdiff.txt: 230 - 250 us for 516 MB): its K tile
// is 64 bytes of a 512-byte row, so every row of dz and y is visited eight times, 64 bytes at a time, and a workgroup's operand ring drains while its row
// loop runs. Here
//   * the weights (w2: 64 channels x 2 x 256 k, 64 KB) are loaded ONCE per workgroup and stay IN REGISTERS: wave (cb, kh) only ever multiplies by its own
//     32 channels x 256 k = 16 KB = 64 VGPRs per lane, exactly the 16 B-operand fragments of its 16 MFMAs (one 4-wave workgroup per CU: 512 VGPRs per lane
//     are there) — which leaves the LDS to the operand ring (NT tiles of 32 pixels, NT - 1 of them in flight while one is multiplied);
//   * a K tile is a WHOLE row: one LDS-DMA instruction moves 2 pixels x 512 contiguous bytes, a stage is 32 pixels of one tensor (16 KB), a tile = the dz
//     stage + the y stage of the same 32 pixels; tiles t + 1 .. t + NT - 1 land while tile t is multiplied AND while its row epilogue runs
//     (the overlap the row-range persistent kernel lacks); the epilogue's own operands (the next BatchNorm's input, the relu' bits) are requested a tile ahead;
//   * 4 waves = 2 channel blocks x 2 K parts (wave (cb, kh) multiplies the 32 pixels by w2[32 cb .., part kh] over 256 k: 16 MFMAs), the two parts meet in
//     LDS, and the 256 threads then each own 8 channels of one pixel: + bias, mask by the packed bits, bf16 store, the two reductions.
// One persistent workgroup per CU walks a contiguous range of pixels. LDS images are [row][512 B] with the 16-byte chunk index XOR-ed by row & 31 (DMA source
// side and fragment read side alike): conflict-free ds_read_b128 fragments.
#include "vec.h"
#include "det.h"
#include "clite.h"
#include "wide_api.h"

using namespace clite;

namespace {

constexpr int FK = 256, FC = 64, FBM = 32;          // k per slot, input channels (= output columns), pixels per tile
constexpr int ROWB = FK * 2;                          // 512-byte rows (bf16)
constexpr int A_STAGE = FBM * ROWB;                   // 16 KB: 32 pixels of one tensor
constexpr int B_PART = FC * ROWB;                     // 32 KB: 64 channels x 256 k of one part
constexpr int STG = FBM * FC * 4;                     // 8 KB: one K part's partial products, f32
#ifndef CLITE_FOLD_ROWS_NT
#define CLITE_FOLD_ROWS_NT 2          // same-box A/B of the captured step (make variant VAR_EXTRA=-DCLITE_FOLD_ROWS_NT=..): 2 tiles (80 KB of LDS) 14.63 / 14.66 ms,
#endif                                 // 4 tiles (144 KB) 14.70 / 14.73: stand-alone the kernel streams at 4.2 TB/s either way, inside the step the smaller
                                       // footprint shares a CU with the text encoder's kernels
constexpr int NT = CLITE_FOLD_ROWS_NT;                // tiles in the ring (2, 3 or 4)
constexpr int SMEM = NT * 2 * A_STAGE + 2 * STG;      // 80 KB at NT = 2

struct FoldArgs {
  const void* pair;          // bf16 [2][M][256]: slot 0 = dz, slot 1 = y
  const void* w2;            // bf16 [64][2][256]: part 0 multiplies y, part 1 dz
  const float* bias;
  void* out;                 // bf16 [M][64]
  const uint8_t* bits;       // [M][8]
  const void* y2;            // bf16 [M][64]: the next BatchNorm's input
  const float* bn_stats;
  int bn_replicas, bn_rstride;
  float bn_inv_count;
  float* colsum;
  int colsum_replicas, colsum_stride;
  int M, tiles, tiles_per_wg;
};

__global__ __launch_bounds__(256) void fold_dgrad_rows_kernel(FoldArgs a) {
  __shared__ __attribute__((aligned(1024))) char smem[SMEM];
  char* aring = smem;
  float* stg = (float*)(aring + NT * 2 * A_STAGE);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_uniform(tid >> 6);
  const int cb = wave & 1, kh = wave >> 1;
  const int T0 = blockIdx.x * a.tiles_per_wg;
  int T1 = T0 + a.tiles_per_wg;
  if (T1 > a.tiles) T1 = a.tiles;
  if (T0 >= T1) return;
  const rsrc_t r_pair = make_rsrc(a.pair, (uint32_t)((size_t)2 * a.M * ROWB));
  const rsrc_t r_w = make_rsrc(a.w2, (uint32_t)(2 * B_PART));
  const rsrc_t r_out = make_rsrc(a.out, (uint32_t)((size_t)a.M * FC * 2));
  const rsrc_t r_y2 = make_rsrc(a.y2, (uint32_t)((size_t)a.M * FC * 2));
  const rsrc_t r_bits = make_rsrc(a.bits, (uint32_t)((size_t)a.M * FC / 8));

  // this lane's slot in a DMA instruction that moves 2 rows x 512 B: row (lane >> 5) of the pair, LDS chunk slot lane & 31; the SOURCE chunk is the slot
  // XOR-ed by the row's low bits (rows 2 i and 2 i + 1 differ in bit 0 only: the XOR of the even row, ^ 1 for the odd one)
  const int lrow = lane >> 5, lslot = lane & 31;

  // ---- the weights, once, into registers: this lane's 16 B-operand fragments (channel 32 cb + (lane & 31), part kh, 16-byte chunk 2 ks + (lane >> 5))
  u32x4 bw[FK / 16];
  {
    const int c = cb * 32 + (lane & 31);
#pragma unroll
    for (int ks = 0; ks < FK / 16; ++ks) bw[ks] = buf_load16(r_w, (uint32_t)((c * 2 + kh) * ROWB + ((2 * ks + (lane >> 5)) << 4)));
  }
  // a tile's two stages: part 0 <- slot 1 (y), part 1 <- slot 0 (dz); 16 instructions per stage, 4 per wave
  auto issue_tile = [&](int T, int tp) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = wave * 4 + j, row = 2 * i + lrow, m = T * FBM + row;
        const uint32_t off = m < a.M ? (uint32_t)(((size_t)(1 - p) * a.M + m) * ROWB) + (uint32_t)((lslot ^ (row & 31)) << 4) : OOB_OFF;
        buf_load16_lds(r_pair, off, aring + (tp * 2 + p) * A_STAGE + 2 * i * ROWB);          // (tp: ring position)
      }
  };
  // the epilogue's thread: pixel erow of the tile, channels ec .. ec + 7
  const int erow = tid >> 3, ec = (tid & 7) * 8;
  float bias[8], mean[8], csum[8], csq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { bias[e] = a.bias[ec + e]; mean[e] = 0.f; csum[e] = 0.f; csq[e] = 0.f; }
  for (int r = 0; r < a.bn_replicas; ++r)
#pragma unroll
    for (int e = 0; e < 8; ++e) mean[e] += a.bn_stats[(size_t)r * a.bn_rstride + ec + e];
#pragma unroll
  for (int e = 0; e < 8; ++e) mean[e] *= a.bn_inv_count;
  auto epi_request = [&](int T, u32x4& yv, uint32_t& bb) {
    const int m = T * FBM + erow;
    const uint32_t gix = (uint32_t)(m * FC + ec);
    yv = buf_load16(r_y2, m < a.M ? gix * 2u : OOB_OFF);
    bb = buf_load1(r_bits, m < a.M ? (gix >> 3) : OOB_OFF);
  };

  // the epilogue operands of the NT tiles in the ring (requested with their tile; rotated by one per iteration)
  u32x4 yq[NT];
  uint32_t bq[NT];
#pragma unroll
  for (int q = 0; q < NT - 1; ++q) {
    // (the epilogue's operands FIRST: whoever waits for them waits for older requests only, not for the tile's 8 younger DMA requests)
    epi_request(T0 + q, yq[q], bq[q]);          // (tiles past T1 address rows >= M or another workgroup's rows: harmless reads, never used)
    if (T0 + q < T1) issue_tile(T0 + q, q);
  }
  // fragment offsets inside a stage (loop-invariant per lane): row r, 16-byte chunk 2 ks + (lane >> 5), XOR-ed by r & 31
  const int fr = lane & 31, fh = lane >> 5;
  int slot = 0;                                  // ring position of tile T

  for (int T = T0; T < T1; ++T) {
    // tiles T + 1 .. T + NT - 2 are in flight; request tile T + NT - 1 into the position tile T - 1 left (behind the previous iteration's barriers)
    int ns = slot + NT - 1; if (ns >= NT) ns -= NT;
    epi_request(T + NT - 1, yq[NT - 1], bq[NT - 1]);
    if (T + NT - 1 < T1) issue_tile(T + NT - 1, ns);
    // everything older than the requests of the tiles behind T has landed: tile T's stages, its operands, the weights. Each tile in flight = 2 + 8 requests
    // of this lane; a tile past T1 issued its 2 epilogue requests only
    {
      const int behind = T1 - 1 - T;             // real tiles behind T
      if (behind >= NT - 1) wait_vmcnt<10 * (NT - 1)>();
      else if (NT > 3 && behind == 2) wait_vmcnt<10 * 2 + 2 * (NT > 3 ? NT - 3 : 0)>();
      else if (NT > 2 && behind == 1) wait_vmcnt<10 * 1 + 2 * (NT > 2 ? NT - 2 : 0)>();
      else wait_vmcnt<2 * (NT - 1)>();
    }
    barrier_raw();
    const char* ast = aring + (slot * 2 + kh) * A_STAGE;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < FK / 16; ++ks) {
      const int c = 2 * ks + fh;
      Chunk16 fa, fb;
      fa.u = *(const u32x4*)(ast + fr * ROWB + ((c ^ fr) << 4));
      fb.u = bw[ks];
      acc = mfma32_bf16(fa.h, fb.h, acc);
    }
    float* sp = stg + kh * (FBM * FC);
#pragma unroll
    for (int r = 0; r < 16; ++r) sp[((r & 3) + 8 * (r >> 2) + 4 * fh) * FC + cb * 32 + fr] = acc[r];
    lds_barrier();
    {
      const float* s0 = stg + erow * FC + ec;
      const f32x4 p0 = *(const f32x4*)s0, p1 = *(const f32x4*)(s0 + 4), q0 = *(const f32x4*)(s0 + FBM * FC), q1 = *(const f32x4*)(s0 + FBM * FC + 4);
      float v[8] = {p0[0] + q0[0], p0[1] + q0[1], p0[2] + q0[2], p0[3] + q0[3], p1[0] + q1[0], p1[1] + q1[1], p1[2] + q1[2], p1[3] + q1[3]};
      Chunk16 yc;
      yc.u = yq[0];
      const uint32_t b_cur = bq[0];
      // rows past the end: zero bits (the buffer bound) -> v = 0, nothing stored (the store offset is out of range), nothing added
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (b_cur >> e) & 1u ? v[e] + bias[e] : 0.f;
      Chunk16 oc;
#pragma unroll
      for (int e = 0; e < 8; ++e) oc.e[e] = f2bf(v[e]);
      const int m = T * FBM + erow;
      buf_store16(r_out, m < a.M ? (uint32_t)(m * FC + ec) * 2u : OOB_OFF, oc.u);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float vs = bf2f(oc.e[e]);          // statistics of what was stored
        csum[e] += vs;
        csq[e] += vs * (bf2f(yc.e[e]) - mean[e]);
      }
    }
#pragma unroll
    for (int q = 0; q + 1 < NT; ++q) { yq[q] = yq[q + 1]; bq[q] = bq[q + 1]; }
    if (++slot == NT) slot = 0;
    // (no barrier here: the next iteration's barrier_raw comes before anything overwrites `stg`, and every wave reaches it behind its reads above)
  }
  // ---- the two reductions: fold the 32 pixel rows of the workgroup through LDS, one atomic per column and statistic
  lds_barrier();
  float* red = stg;                                  // [32][8][16]
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[(erow * 8 + (tid & 7)) * 16 + e] = csum[e];
    red[(erow * 8 + (tid & 7)) * 16 + 8 + e] = csq[e];
  }
  lds_barrier();
  if (tid < 128 && a.colsum) {
    float s = 0.f;
    for (int r = 0; r < 32; ++r) s += red[r * 128 + tid];
    const int chunk = tid >> 4, e = tid & 15;
    float* crep = a.colsum + (a.colsum_replicas > 1 ? (size_t)(blockIdx.x % a.colsum_replicas) * a.colsum_stride : 0);
    atomic_add_f32(crep + (e >= 8 ? FC : 0) + chunk * 8 + (e & 7), s);
  }
}

}  // namespace

#ifndef CLITE_FOLD_ROWS_WGS
#define CLITE_FOLD_ROWS_WGS 256          // one persistent workgroup per CU (144 KB of LDS each); the wave-simulator build of the tests sets 3
#endif

int clite::launch_fold_dgrad_rows(const void* pair, const void* w2, int M, int K, int Cin, const clite_epilogue& ep, hipStream_t st) {
#ifdef CLITE_NO_FOLD_ROWS
  return WIDE_NOT_TAKEN;
#endif
  if (K != FK || Cin != FC || deterministic() || tile_policy_value() != 0) return WIDE_NOT_TAKEN;
  if (!ep.bias || !ep.relu_bits || !ep.bn_y || !ep.bn_stats || !ep.colsum || ep.colsum_rows == 1 || ep.out_f32 || ep.ldc != FC || ep.alpha != 1.f || ep.residual || ep.dact_aux ||
      ep.mask_after_residual || ep.atomic || ep.act || ep.preact || ep.drop_p > 0.f)
    return WIDE_NOT_TAKEN;
  if ((size_t)2 * M * ROWB >= 0xF0000000ull) return WIDE_NOT_TAKEN;
  FoldArgs a;
  a.pair = pair; a.w2 = w2; a.bias = ep.bias; a.out = ep.out; a.bits = ep.relu_bits; a.y2 = ep.bn_y; a.bn_stats = ep.bn_stats;
  a.bn_replicas = ep.bn_replicas; a.bn_rstride = ep.bn_rstride; a.bn_inv_count = ep.bn_inv_count;
  a.colsum = ep.colsum; a.colsum_replicas = ep.colsum_replicas; a.colsum_stride = ep.colsum_stride;
  a.M = M;
  a.tiles = (M + FBM - 1) / FBM;
  int wgs = a.tiles < CLITE_FOLD_ROWS_WGS ? a.tiles : CLITE_FOLD_ROWS_WGS;
  a.tiles_per_wg = (a.tiles + wgs - 1) / wgs;
  wgs = (a.tiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
  hipLaunchKernelGGL(fold_dgrad_rows_kernel, dim3(wgs), dim3(256), 0, st, a);
  return (int)hipGetLastError();
}
