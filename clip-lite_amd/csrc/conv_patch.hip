// Patch-resident 3 x 3 / stride 1 / pad 1 convolution for 64 -> 64 channels (bf16): torchvision ResNet's layer1 conv2 (reference encoder.py:36-41;
// block arithmetic as restated in model_zoo/resnet.py:60-100) — forward, and its input gradient on the transposed weight copy.
//
// Why its own kernel (round 4; DESIGN.md §3.1, "patch-resident"). As an implicit GEMM (M = pixels, N = 64, K = 9 x 64) the tile engine gathers
// every input pixel NINE times out of L2 — 288 KB of gathered operand per 256-pixel tile whose input patch with halo is 44 KB — and at 56 x 56
// that gather, not HBM and not the matrix pipe, is the launch: 70.6 us forward / 83.3 us input gradient against an HBM ideal of 19 / 29
// (profiles/r3_layers.txt). Here a workgroup keeps in LDS
//   * ALL nine taps of the weights, [tap][out channel][64 in channels] = 72 KB, loaded once per workgroup, and
//   * the input PATCH of a strip of R output rows: (R + 2) x (W + 1) pixels x 128 B (R = 4 at W = 56: 43 KB), double-buffered,
// and forms every tap's A fragment by a `ds_read_b128` at a shifted patch address: L2 -> LDS traffic per tile drops 6.5 x (patch once + nothing
// for the weights), HBM traffic stays the algorithmic input + output. One persistent 8-wave workgroup per CU walks a contiguous range of strips
// (1792 strips at batch 128 = 7 per CU); strip i + 1's patch lands by LDS-DMA while strip i is multiplied and written.
//
// Patch image. Pixel (pr, pc) of the patch — pr = 0 .. R + 1 (image row y0 - 1 + pr), pc = 0 .. W + 1 (image column pc - 1) — sits at flat index
// pr * (W + 1) + pc: the right halo of row pr IS the left halo of row pr + 1 (both are zeros), so a row costs W + 1 pixels, not W + 2. Halo and
// out-of-image pixels are zero-filled by out-of-range DMA sources. The 16-byte chunk c of pixel p sits in slot c ^ ((p >> 1) & 7): 16 lanes reading
// chunk c of 16 consecutive pixels then cover all 64 banks once (pixel pitch 128 B = 32 banks; a strip-row crossing inside a lane group skips one
// pixel and costs a 2-way conflict on one pair). The DMA destination is lane-linear, so the swizzle is applied to the SOURCE chunk (igemm_dma.h).
//
// Waves: 4 (M) x 2 (N), each 64 pixels x 32 output channels = 2 MFMA 32 x 32 x 16 blocks; a wave skips the blocks beyond the strip (224 pixels at
// W = 56: 7 of 8 blocks). Per k-step a wave reads 2 A + 1 B fragments for 2 MFMAs: 192 B/clk/CU of LDS reads at full matrix rate.
//
// Epilogue forms: 0 = what clite_conv_fwd's launches ask for (bf16 store + per-channel sum / sum of squares of the stored values); 1 = the
// BatchNorm-backward form of the ResNet backward's dgrads (packed relu' bits, bf16 store, sum v and sum v (bn_y - mean): igemm.h FORM 1). The
// accumulators go through the strip's own (now free) patch buffer in two halves of 128 rows; stores are whole 128-byte pixel rows.
#include "igemm_wide.h"
#include "wide_api.h"
#include <type_traits>
#include "stem_bn.h"

using namespace clite;

#ifndef CLITE_PATCH_WGS
#define CLITE_PATCH_WGS 256          // one persistent workgroup per CU (the simulator build uses 2, so that its small cases walk several strips)
#endif

namespace {

constexpr int PC = 64;                        // channels in = channels out
constexpr int PIXB = PC * 2;                  // bytes per pixel
constexpr int NTAP = 9;
constexpr int WBYTES = NTAP * PC * PIXB;      // 73,728: [tap][out channel][in channel]
constexpr int MAXPIX = 344;                   // (R + 2) (W + 1) + 1 <= 344
constexpr int PBYTES = MAXPIX * PIXB;         // 44,032 per patch buffer
constexpr int NPI = MAXPIX / 8;               // DMA instructions per patch (8 pixels each)
constexpr int NPW = (NPI + 7) / 8;            // ... per wave
constexpr int EPITCH = PC * 4 + 16;           // f32 staging row
constexpr int HALF = 128;                     // rows staged at a time
static_assert(HALF * EPITCH <= PBYTES, "a half of the accumulator tile fits a patch buffer");
static_assert(WBYTES + 2 * PBYTES <= 160 * 1024, "LDS budget");

struct PatchArgs {
  const void* x;
  uint32_t xbytes;
  const void* w;
  uint32_t wbytes;
  int N, H, W, R;             // R: output rows per strip
  int strips_per_img, nstrips;
  int flip;                   // 1: tap t reads the weight's tap 8 - t (input gradient on [C][R][S][K] weights)
};

#ifndef CLITE_PATCH_NT
#define CLITE_PATCH_NT 1              // 1: the strips' input patches / dy rows are loaded with the non-temporal policy (each byte is read by one CU, once or twice)
#endif
#if CLITE_PATCH_NT
#define PATCH_LOAD buf_load16_lds_nt
#else
#define PATCH_LOAD buf_load16_lds
#endif
#ifndef CLITE_PATCH_ABLATE
#define CLITE_PATCH_ABLATE 0          // diagnostic variant builds only (tools/probe_patch.py): 1 = no MFMA loop, 2 = no global stores, 4 = no patch DMA after the first
#endif

template <int FORM>
__global__ __launch_bounds__(512) void conv3x3_patch_kernel(PatchArgs a, Epilogue ep, int strips_per_wg) {
  __shared__ __attribute__((aligned(1024))) char smem[WBYTES + 2 * PBYTES];
  char* const wbuf = smem;
  char* const pbuf = smem + WBYTES;
  typedef bf16 T;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = wave_uniform(tid >> 6);
  const int wmi = wave >> 1, wni = wave & 1;
  const int W1 = a.W + 1;
  const int rows_strip = a.R * a.W;                         // GEMM rows of a full strip (<= 256)

  int s_begin = blockIdx.x * strips_per_wg, s_end = s_begin + strips_per_wg;
  if (s_end > a.nstrips) s_end = a.nstrips;
  if (s_begin >= s_end) return;

  // ---- the weights, once: 72 DMA instructions of 8 rows x 128 B; row = tap * 64 + out channel
  {
    const rsrc_t rw = make_rsrc(a.w, a.wbytes);
#pragma unroll
    for (int k = 0; k < NTAP; ++k) {
      const int i = wave + 8 * k;
      const int row = i * 8 + (lane >> 3);
      const int t = row >> 6, oc = row & 63;
      const int ts = a.flip ? 8 - t : t;
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      buf_load16_lds(rw, (uint32_t)(((oc * NTAP + ts) * PC + chunk * 8) * 2), wbuf + i * 1024);
    }
  }

  // ---- patch DMA: per-lane constants of this wave's instructions (the same for every strip)
  const rsrc_t rx = make_rsrc(a.x, a.xbytes);
  int p_rel[NPW], p_row[NPW];          // element offset relative to the strip's first pixel; patch row (-1: never valid)
#pragma unroll
  for (int k = 0; k < NPW; ++k) {
    const int i = wave + 8 * k;
    const int pix = i * 8 + (lane >> 3);
    const int pr = pix / W1, pc = pix - pr * W1;
    const int chunk = (lane & 7) ^ ((pix >> 1) & 7);
    const bool ok = i < NPI && pc >= 1 && pc <= a.W && pr <= a.R + 1;
    p_row[k] = ok ? pr : -1;
    p_rel[k] = ((pr - 1) * a.W + (pc - 1)) * PC + chunk * 8;
  }
  auto issue_patch = [&](int strip, char* dst) {
    const int n = strip / a.strips_per_img;
    const int y0 = (strip - n * a.strips_per_img) * a.R;
    const int base = (n * a.H + y0) * a.W * PC;
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
      const int i = wave + 8 * k;
      if (i < NPI) {          // wave-uniform
        const bool v = p_row[k] >= 0 && (unsigned)(y0 + p_row[k] - 1) < (unsigned)a.H;
        PATCH_LOAD(rx, v ? (uint32_t)((base + p_rel[k]) * 2) : OOB_OFF, dst + i * 1024);
      }
    }
  };

  // ---- K loop constants
  int nblk = (rows_strip - 64 * wmi + 31) / 32;
  nblk = nblk < 0 ? 0 : (nblk > 2 ? 2 : nblk);
  nblk = wave_uniform(nblk);
  int bpix[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    int q = 64 * wmi + 32 * mb + (lane & 31);
    if (q > rows_strip - 1) q = rows_strip - 1;
    const int oy = q / a.W, ox = q - oy * a.W;
    bpix[mb] = oy * W1 + ox;
  }
  const int cl = lane >> 5;
  const int rb = 32 * wni + (lane & 31);
  const int swb = (rb >> 1) & 7;
  int bsw[4];
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) bsw[kc] = rb * PIXB + (((kc * 2 + cl) ^ swb) << 4);

  // ---- epilogue constants
  const int ecol = (tid & 7) * 8, erow0 = tid >> 3;          // 64 rows per sweep of the 512 threads
  float csum[8], csq[8], bn_mean[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { csum[e] = 0.f; csq[e] = 0.f; bn_mean[e] = 0.f; }
  if (FORM == 1) {
    for (int r = 0; r < ep.bn_replicas; ++r)
#pragma unroll
      for (int e = 0; e < 8; ++e) bn_mean[e] += ep.bn_stats[(size_t)r * ep.bn_rstride + ecol + e];
#pragma unroll
    for (int e = 0; e < 8; ++e) bn_mean[e] *= ep.bn_inv_count;
  }
  const bool stats = ep.colsum != nullptr;
  const rsrc_t r_out = make_rsrc(ep.out, RSRC_WHOLE), r_y = make_rsrc(ep.bn_y, RSRC_WHOLE), r_bits = make_rsrc(ep.relu_bits, RSRC_WHOLE);

  issue_patch(s_begin, pbuf);
  wait_vmcnt<0>();
  lds_barrier();

  for (int strip = s_begin; strip < s_end; ++strip) {
    const int cur = (strip - s_begin) & 1;
    char* const pcur = pbuf + cur * PBYTES;
    if (strip + 1 < s_end && !(CLITE_PATCH_ABLATE & 4)) issue_patch(strip + 1, pbuf + (cur ^ 1) * PBYTES);
    const int n = strip / a.strips_per_img;
    const int y0 = (strip - n * a.strips_per_img) * a.R;
    int rows_valid = (a.H - y0 < a.R ? a.H - y0 : a.R) * a.W;
    const uint32_t row0 = (uint32_t)((n * a.H + y0) * a.W);
    // FORM 1: this strip's epilogue operands (relu' bits, BatchNorm input) are requested now and land during the K loop
    uint32_t pb[4];
    Raw8<T> py[4];
    if (FORM == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int trow = (j >> 1) * HALF + erow0 + 64 * (j & 1);
        const uint32_t gix = (row0 + trow) * PC + ecol;
        const bool v = trow < rows_valid;
        pb[j] = buf_load1(r_bits, v ? (gix >> 3) : OOB_OFF);
        py[j].ldb(r_y, v ? gix * 2u : OOB_OFF);
      }
    }

    f32x16 acc[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
    // 36 k-steps (9 taps x 4 chunks of 16 channels) of straight-line code per block count, the next k-step's three fragment reads issued ahead of
    // this k-step's MFMAs. (A block-count test INSIDE the loop made the compiler emit read -> s_waitcnt lgkmcnt(0) -> MFMA per fragment, every
    // MFMA behind a full LDS round trip: 46 us per launch instead of ~30.)
    // (the 72 fragment addresses are functions of per-lane constants: left alone, the compiler hoists them all out of the strip loop — 232 live
    // registers, spills in the BatchNorm-backward form; made opaque here they are three VALU operations next to each read, in the MFMAs' shadow)
    int bp[2] = {bpix[0], bpix[1]};
    opaque_i(bp[0]);
    opaque_i(bp[1]);
    auto kloop = [&](auto nb_tag) {
      constexpr int NB = decltype(nb_tag)::value;
      auto a_addr = [&](int ks, int mb) -> const char* {
        const int t = ks >> 2, kc = ks & 3;
        const int pix = bp[mb] + (t / 3) * W1 + (t % 3);
        return pcur + pix * PIXB + ((((kc * 2 + cl) ^ (pix >> 1)) & 7) << 4);
      };
      auto b_addr = [&](int ks) -> const char* { return wbuf + (ks >> 2) * (PC * PIXB) + bsw[ks & 3]; };
      Chunk16 fa[2][2], fb[2];
      fb[0].u = *(const u32x4*)b_addr(0);
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) fa[0][mb].u = *(const u32x4*)a_addr(0, mb);
#pragma unroll
      for (int ks = 0; ks < NTAP * 4; ++ks) {
        const int c = ks & 1, nx = c ^ 1;
        if (ks + 1 < NTAP * 4) {
          fb[nx].u = *(const u32x4*)b_addr(ks + 1);
#pragma unroll
          for (int mb = 0; mb < NB; ++mb) fa[nx][mb].u = *(const u32x4*)a_addr(ks + 1, mb);
        }
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) acc[mb] = mfma32_bf16(fa[c][mb].h, fb[c].h, acc[mb]);
      }
    };
    if (!(CLITE_PATCH_ABLATE & 1)) {
      if (nblk > 1) kloop(std::integral_constant<int, 2>{});
      else if (nblk == 1) kloop(std::integral_constant<int, 1>{});
    }
    wait_vmcnt<0>();          // the next strip's patch and this strip's epilogue operands have landed (they had the K loop to do so)
    lds_barrier();            // every wave is past its last read of this patch: the buffer becomes the accumulator staging

#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if ((wmi >> 1) == h) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = 64 * (wmi & 1) + 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int col = 32 * wni + (lane & 31);
            *(float*)(pcur + row * EPITCH + col * 4) = acc[mb][r];
          }
      }
      lds_barrier();
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int rr = erow0 + 64 * k;
        const int trow = h * HALF + rr;
        const bool valid = trow < rows_valid;
        const float* src = (const float*)(pcur + rr * EPITCH + ecol * 4);
        const f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        const uint32_t gix = (row0 + trow) * PC + ecol;
        if (FORM == 1) {
          const uint32_t bits = pb[h * 2 + k];          // zeros for a row past the strip
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (bits >> e) & 1u ? v[e] : 0.f;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = valid ? v[e] : 0.f;
        }
        Chunk16 ch;
#pragma unroll
        for (int e = 0; e < 8; ++e) ch.e[e] = f2bf(v[e]);
        buf_store16(r_out, (valid && !(CLITE_PATCH_ABLATE & 2)) ? gix * 2u : OOB_OFF, ch.u);
        if (FORM == 1) {
          float yv[8];
          py[h * 2 + k].get(yv);
          round8_bf16(v);
#pragma unroll
          for (int e = 0; e < 8; ++e) { csum[e] += v[e]; csq[e] += v[e] * (yv[e] - bn_mean[e]); }
        } else if (stats) {
          round8_bf16(v);   // statistics of what was stored
#pragma unroll
          for (int e = 0; e < 8; ++e) { csum[e] += v[e]; csq[e] += v[e] * v[e]; }
        }
      }
      lds_barrier();
    }
  }

  if (stats) {
    // one flush per workgroup: the 64 threads that share a column chunk fold through LDS (any patch buffer is free now)
    float* red = (float*)pbuf;                      // [64 rows][8 chunks x 16]
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[erow0 * 128 + (tid & 7) * 16 + e] = csum[e];
      red[erow0 * 128 + (tid & 7) * 16 + 8 + e] = csq[e];
    }
    lds_barrier();
    if (tid < 128) {
      float s = 0.f;
      for (int r = 0; r < 64; ++r) s += red[r * 128 + tid];
      const int chunk = tid >> 4, e = tid & 15;
      const int col = chunk * 8 + (e & 7);
      float* crep = ep.colsum + (ep.colsum_replicas > 1 ? (size_t)(blockIdx.x % ep.colsum_replicas) * ep.colsum_stride : 0);
      if (e < 8 || ep.colsum_rows != 1) atomic_add_f32(crep + (e >= 8 ? PC : 0) + col, s);
    }
  }
}

}  // namespace

// Returns WIDE_NOT_TAKEN when the launch is not one this kernel covers (the caller then takes the implicit-GEMM path).
int clite::launch_conv3x3_patch(const void* x, const void* w, const clite_conv& c, const clite_epilogue& ep, bool dgrad, hipStream_t st) {
#ifdef CLITE_NO_PATCH          // A/B builds only (tools/build_patch_variants.sh): every launch on the implicit-GEMM path
  return WIDE_NOT_TAKEN;
#endif
  if (c.dtype != CLITE_BF16 || c.C != PC || c.K != PC || c.R != 3 || c.S != 3 || c.stride != 1 || c.pad != 1 || c.Ho != c.H || c.Wo != c.W) return WIDE_NOT_TAKEN;
  if (ep.ldc != PC || ep.atomic || ep.out_f32 || ep.alpha != 1.f || ep.bias || ep.act != CLITE_ACT_NONE || ep.preact || ep.dact_aux || ep.drop_p > 0.f ||
      ep.residual || ep.mask_after_residual || ep.splitk_ws)
    return WIDE_NOT_TAKEN;
  int form;
  if (!ep.bn_y && !ep.relu_bits) form = 0;
  else if (ep.bn_y && ep.relu_bits && ep.colsum) form = 1;
  else return WIDE_NOT_TAKEN;
  if (!dgrad && form != 0) return WIDE_NOT_TAKEN;
  int R = 256 / c.W;
  if (R > c.H) R = c.H;
  while (R >= 1 && (R + 2) * (c.W + 1) + 1 > MAXPIX) --R;
  if (R < 1) return WIDE_NOT_TAKEN;
  PatchArgs a;
  a.x = x; a.xbytes = (uint32_t)((size_t)c.N * c.H * c.W * PC * 2);
  a.w = w; a.wbytes = (uint32_t)(NTAP * PC * PC * 2);
  a.N = c.N; a.H = c.H; a.W = c.W; a.R = R;
  a.strips_per_img = (c.H + R - 1) / R;
  a.nstrips = c.N * a.strips_per_img;
  a.flip = dgrad ? 1 : 0;
  const int grid = a.nstrips < CLITE_PATCH_WGS ? a.nstrips : CLITE_PATCH_WGS;
  const int per = (a.nstrips + grid - 1) / grid;
  const int g2 = (a.nstrips + per - 1) / per;
  if (form == 0) hipLaunchKernelGGL(conv3x3_patch_kernel<0>, dim3(g2), dim3(512), 0, st, a, ep, per);
  else hipLaunchKernelGGL(conv3x3_patch_kernel<1>, dim3(g2), dim3(512), 0, st, a, ep, per);
  return (int)hipGetLastError();
}

// =====================================================================================================================================
// Patch-resident WEIGHT GRADIENT of the same convolution: dW[co][r][s][ci] (f32) += sum over pixels of dy[p][co] * x[p + (r - 1, s - 1)][ci].
//
// In the grouped implicit-GEMM form a 64 -> 64 3 x 3 member at 56 x 56 takes 97 us for an HBM ideal of 19: its output is tiny (64 x 576) and its
// contraction enormous (401 408 pixels), every workgroup owns a [64][128-column] output tile, so dy is streamed once per column tile (5 x) and
// every input pixel is gathered nine times (tools/probe_wgrad_s2.py). Here ONE workgroup owns the WHOLE 64 x 9 x 64 output
// in registers (36 MFMA blocks over 8 waves: waves 0-3 a tap each plus a quarter of tap 8, waves 4-7 a tap each) and walks a contiguous range
// of strips; per strip the dy rows and the input patch land ONCE in LDS (double-buffered LDS-DMA) and all nine taps read the patch at shifted
// addresses. Both operands are contracted over the PIXEL index, so both LDS images are [pixel][64 channels] and every fragment is a pair of
// `ds_read_b64_tr_b16` transposed reads (igemm_dma.h's XC image).
//
// Flat pixel index. dy pixel (oy, ox) of the strip sits at g = oy (W + 1) + ox; the entries with ox = W, and everything past the strip, are ZERO
// (out-of-range DMA sources), so the contraction simply runs over all flat indices 0 .. 16 ceil(R (W + 1) / 16) - 1: a zero dy contributes
// nothing. The input patch uses conv3x3_patch_kernel's flat layout, in which tap (r, s) of pixel g is patch entry g + r (W + 1) + s — one shared
// zero column serves as the right halo of a row and the left halo of the next.
// Swizzle: the 64-byte half h of pixel p sits in half h ^ ((p >> 1) & 1): the four consecutive pixels a 16-lane group of a transposed read touches
// then cover four different 16-bank groups whatever the tap's shift (p and p + 4 swizzle alike, as the second read of a fragment needs).
//
// The partial sums of a workgroup (36 864 floats) go to its slab of a workspace; wgrad_patch_reduce_kernel adds the slabs into dW. (Float atomics
// straight into dW would be 37.7 MB of atomic traffic at ~1.3 TB/s: 29 us; slabs + reduction move the same bytes at HBM rate.)
namespace {

constexpr int WG_DYPIX = 256;                       // flat dy entries per strip = 16 k-steps of 16 (R (W + 1) <= 256; the K loop runs an even number of k-steps)
constexpr int WG_XPIX = 376;                        // patch entries reachable: 255 + 2 (W + 1) + 2 <= 375 for W <= 58
constexpr int WG_DYB = WG_DYPIX * PIXB, WG_XB = WG_XPIX * PIXB, WG_STAGE = WG_DYB + WG_XB;
constexpr int WG_NI_DY = WG_DYPIX / 8, WG_NI_X = WG_XPIX / 8;             // DMA instructions per strip
constexpr int WG_NW_DY = (WG_NI_DY + 7) / 8, WG_NW_X = (WG_NI_X + 7) / 8;  // ... per wave
constexpr int WG_OUT = NTAP * PC * PC;              // 36 864 partial sums per workgroup
static_assert(2 * WG_STAGE <= 160 * 1024, "LDS budget");

struct WgradPatchArgs {
  const void* dy;
  const void* x;
  uint32_t bytes;             // of either tensor
  int N, H, W, R;
  int strips_per_img, nstrips;
  float* ws;                  // [gridDim.x][9][64][64] f32 slabs
};

// Fragment of a [pixel][64 ch] image: 32 channels starting at x0, 16 pixels starting at pixel pix0 + k0 (igemm_dma.h DmaXCStrided::frag_off /
// frag_at with this kernel's swizzle). For k0 a multiple of 4 the swizzle bit of pixel pix0 + k0 + j equals that of pix0 + j, so the byte offset
// is wg_frag_base(pix0, ...) + k0 * 128: the K loop adds a compile-time immediate to a per-lane base and spends no VALU on addresses (with them
// computed per fragment the loop was issue-bound: ~60 VALU + 36 LDS instructions per k-step and SIMD against 9 MFMAs).
DEV int wg_frag_base(int pix0, int x0, int lane) {
  const int x = x0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int pix = pix0 + 8 * (lane >> 5) + ((lane >> 2) & 3);
  const int b = x * 2;
  return pix * PIXB + ((((b >> 6) ^ (pix >> 1)) & 1) << 6) + (b & 63);
}
DEV bf16x8 wg_frag_at(const char* p) {
  const s16x4 lo = lds_read_tr16(p);
  const s16x4 hi = lds_read_tr16(p + 4 * PIXB);
  union { s16x4 v[2]; bf16x8 h; } u;
  u.v[0] = lo; u.v[1] = hi;
  return u.h;
}

__global__ __launch_bounds__(512) void conv3x3_wgrad_patch_kernel(WgradPatchArgs a, int strips_per_wg) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * WG_STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = wave_uniform(tid >> 6);
  const int W1 = a.W + 1;

  int s_begin = blockIdx.x * strips_per_wg, s_end = s_begin + strips_per_wg;
  if (s_end > a.nstrips) s_end = a.nstrips;

  const rsrc_t rdy = make_rsrc(a.dy, a.bytes), rx = make_rsrc(a.x, a.bytes);
  // per-lane constants of this wave's DMA instructions
  int d_rel[WG_NW_DY], d_row[WG_NW_DY], x_rel[WG_NW_X], x_row[WG_NW_X];
#pragma unroll
  for (int k = 0; k < WG_NW_DY; ++k) {
    const int i = wave + 8 * k;
    const int pix = i * 8 + (lane >> 3);
    const int oy = pix / W1, ox = pix - oy * W1;
    const int chunk = (lane & 7) ^ (((pix >> 1) & 1) << 2);
    const bool ok = i < WG_NI_DY && ox < a.W && oy < a.R;
    d_row[k] = ok ? oy : -1;
    d_rel[k] = (oy * a.W + ox) * PC + chunk * 8;
  }
#pragma unroll
  for (int k = 0; k < WG_NW_X; ++k) {
    const int i = wave + 8 * k;
    const int pix = i * 8 + (lane >> 3);
    const int pr = pix / W1, pc = pix - pr * W1;
    const int chunk = (lane & 7) ^ (((pix >> 1) & 1) << 2);
    const bool ok = i < WG_NI_X && pc >= 1 && pc <= a.W && pr <= a.R + 1;
    x_row[k] = ok ? pr : -1;
    x_rel[k] = ((pr - 1) * a.W + (pc - 1)) * PC + chunk * 8;
  }
  auto issue = [&](int strip, char* dst) {
    const int n = strip / a.strips_per_img;
    const int y0 = (strip - n * a.strips_per_img) * a.R;
    const int base = (n * a.H + y0) * a.W * PC;
#pragma unroll
    for (int k = 0; k < WG_NW_DY; ++k) {
      const int i = wave + 8 * k;
      if (i < WG_NI_DY) {
        const bool v = d_row[k] >= 0 && y0 + d_row[k] < a.H;
        PATCH_LOAD(rdy, v ? (uint32_t)((base + d_rel[k]) * 2) : OOB_OFF, dst + i * 1024);
      }
    }
#pragma unroll
    for (int k = 0; k < WG_NW_X; ++k) {
      const int i = wave + 8 * k;
      if (i < WG_NI_X) {
        const bool v = x_row[k] >= 0 && (unsigned)(y0 + x_row[k] - 1) < (unsigned)a.H;
        PATCH_LOAD(rx, v ? (uint32_t)((base + x_rel[k]) * 2) : OOB_OFF, dst + WG_DYB + i * 1024);
      }
    }
  };

  // this wave's blocks: its own tap (2 x 2 blocks of 32 x 32) and, for waves 0-3, block (wave >> 1, wave & 1) of tap 8
  const int tap = wave;
  const int shift = (tap / 3) * W1 + (tap % 3);
  const int shift8 = 2 * W1 + 2;
  const bool extra = wave < 4;
  // per-lane byte offsets of the five fragments at k-step 0 (dy: both channel halves; x: this wave's tap, both halves; tap 8: one half)
  const int oa0 = wg_frag_base(0, 0, lane), oa1 = wg_frag_base(0, 32, lane);
  const int ob0 = wg_frag_base(shift, 0, lane), ob1 = wg_frag_base(shift, 32, lane), o8 = wg_frag_base(shift8, 32 * (wave & 1), lane);
  f32x16 acc[2][2], acc8;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][0][r] = 0.f; acc[0][1][r] = 0.f; acc[1][0][r] = 0.f; acc[1][1][r] = 0.f; acc8[r] = 0.f; }

  if (s_begin < s_end) issue(s_begin, smem);
  for (int strip = s_begin; strip < s_end; ++strip) {
    const int cur = (strip - s_begin) & 1;
    wait_vmcnt<0>();
    lds_barrier();          // strip's images landed (every wave's part); every wave is past the K loop that read the other stage
    if (strip + 1 < s_end) issue(strip + 1, smem + (cur ^ 1) * WG_STAGE);
    const char* dyi = smem + cur * WG_STAGE;
    const char* xi = dyi + WG_DYB;
    // 16 k-steps of 16 flat pixels, always (the dy image is zero past the strip), fully unrolled: immediate offsets, the next k-step's fragments
    // requested ahead of this one's MFMAs; compile-time block count per wave class (no wave-uniform branch between reads and MFMAs)
    auto kloop = [&](auto extra_tag) {
      constexpr bool EXTRA = decltype(extra_tag)::value;
      const char* pa0 = dyi + oa0; const char* pa1 = dyi + oa1;
      const char* pb0 = xi + ob0; const char* pb1 = xi + ob1; const char* p8 = xi + o8;
      bf16x8 fa[2][2], fb[2][2], f8[2];
      auto load = [&](int set, int ks) {
        fa[set][0] = wg_frag_at(pa0 + ks * 16 * PIXB); fa[set][1] = wg_frag_at(pa1 + ks * 16 * PIXB);
        fb[set][0] = wg_frag_at(pb0 + ks * 16 * PIXB); fb[set][1] = wg_frag_at(pb1 + ks * 16 * PIXB);
        if (EXTRA) f8[set] = wg_frag_at(p8 + ks * 16 * PIXB);
      };
      load(0, 0);
#pragma unroll
      for (int ks = 0; ks < WG_DYPIX / 16; ++ks) {
        const int c = ks & 1;
        if (ks + 1 < WG_DYPIX / 16) load(c ^ 1, ks + 1);
        acc[0][0] = mfma32_bf16(fa[c][0], fb[c][0], acc[0][0]);
        acc[0][1] = mfma32_bf16(fa[c][0], fb[c][1], acc[0][1]);
        acc[1][0] = mfma32_bf16(fa[c][1], fb[c][0], acc[1][0]);
        acc[1][1] = mfma32_bf16(fa[c][1], fb[c][1], acc[1][1]);
        if (EXTRA) acc8 = mfma32_bf16((wave >> 1) ? fa[c][1] : fa[c][0], f8[c], acc8);
      }
    };
    if (extra) kloop(std::true_type{}); else kloop(std::false_type{});
  }

  // partial sums -> this workgroup's slab [tap][co][ci]; lane l holds D[co = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][ci = l & 31] of a block
  float* slab = a.ws + (size_t)blockIdx.x * WG_OUT;
  auto put = [&](const f32x16& v, int t, int cob, int cib) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cob * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      slab[(t * PC + co) * PC + cib * 32 + (lane & 31)] = v[r];
    }
  };
  put(acc[0][0], tap, 0, 0); put(acc[0][1], tap, 0, 1); put(acc[1][0], tap, 1, 0); put(acc[1][1], tap, 1, 1);
  if (extra) put(acc8, 8, wave >> 1, wave & 1);
}

// dw[co][t][ci] += sum over slabs of ws[b][t][co][ci]; grid (144, 16): each workgroup sums a sixteenth of the slabs for 256 outputs (four independent
// chains per thread: the loop is load latency, not arithmetic)
__global__ __launch_bounds__(256) void wgrad_patch_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nslabs) {
  const int i = blockIdx.x * 256 + threadIdx.x;          // index into [t][co][ci]
  const int per = (nslabs + gridDim.y - 1) / gridDim.y;
  int b0 = blockIdx.y * per, b1 = b0 + per;
  if (b1 > nslabs) b1 = nslabs;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = b0;
  for (; b + 3 < b1; b += 4) {
    s0 += ws[(size_t)b * WG_OUT + i]; s1 += ws[(size_t)(b + 1) * WG_OUT + i]; s2 += ws[(size_t)(b + 2) * WG_OUT + i]; s3 += ws[(size_t)(b + 3) * WG_OUT + i];
  }
  for (; b < b1; ++b) s0 += ws[(size_t)b * WG_OUT + i];
  const float s = (s0 + s1) + (s2 + s3);
  const int t = i / (PC * PC), co = (i / PC) % PC, ci = i % PC;
  if (b0 < b1) atomic_add_f32(dw + (co * NTAP + t) * PC + ci, s);
}

}  // namespace

// =====================================================================================================================================
// Patch-resident weight gradient of the STEM (7 x 7 / stride 2 on the pre-padded NHWC4 image, gemm.hip: clite_stem_fwd): dw[c][r][s][ch] (f32) +=
// sum over output pixels of dy[p][c] * xpad[2 oy + r][2 ox + s][ch]. As an implicit GEMM it has 64 x 224 outputs and 1.6 M pixels to contract; the
// grouped form streams dy (205 MB at batch 128) and gathers every input pixel ~12 times out of L2 (189 us for an HBM ideal of 47). Here, as in
// conv3x3_wgrad_patch_kernel, ONE workgroup owns the whole output in registers and walks a contiguous range of strips of two output rows:
//   dy strip    2 Wo pixels x 128 B, contiguous in memory, LDS-DMA into the swizzled [pixel][64 ch] image (A fragments: wg_frag_base / wg_frag_at)
//   x  patch    input rows 2 oy0 .. 2 oy0 + 8: ONE contiguous block of 9 Wp x 8 B, copied as it lies. For a fixed tap row r the 32 columns (s, ch) of
//               pixel ox are the 64 contiguous bytes at ((2 oyl + r) Wp + 2 ox) x 8: a [pixel][32] image whose rows start 16 B apart and overlap,
//               which a transposed read does not mind (every lane supplies its own row address).
// 8 waves: wave = (kh, rp); it multiplies the k-steps (16 pixels) of parity kh for tap rows 2 rp, 2 rp + 1 (rp = 3: row 6 only) and both channel
// halves: 4 (2) accumulator blocks. The two parities meet in LDS at the end; the workgroup's [64][224] partial sums go to its slab of the workspace
// and stem_wgrad_reduce_kernel adds the slabs straight into the [64][7][7][3] gradient (dropping the s = 7 / ch = 3 padding columns).
namespace {

constexpr int SW_MAXPIX = 256;                      // dy pixels per strip (2 Wo <= 256)
constexpr int SW_DYB = SW_MAXPIX * PIXB;            // 32 KB
constexpr int SW_XB = 20 * 1024;                    // 10 Wp x 8 B + 128 <= 20 KB: Wp <= 254
constexpr int SW_STAGE = SW_DYB + SW_XB;
constexpr int SW_OUT = 64 * 224;
static_assert(2 * SW_STAGE <= 160 * 1024 && SW_OUT * 4 <= 2 * SW_STAGE && 3 * SW_STAGE <= 160 * 1024, "LDS budget");

struct StemWgradArgs {
  const void* dy;
  const void* x;
  uint32_t dybytes, xbytes;
  int N, Hp, Wp, Ho, Wo;
  int strips_per_img, nstrips;          // strips of 2 output rows
  float* ws;                            // [gridDim.x][64][224] f32 slabs
};

// NW = 8: one workgroup per CU, two LDS stages (the next strip's images land under this one's K loop), the k-steps split between two wave sets.
// NW = 4 (default): THREE single-stage workgroups per CU - a persistent workgroup with one strip of loads in flight runs in lockstep with the other
// 255 and sees the DMA round trip per strip (133 us at batch 128: 2.4 TB/s); three independent workgroups per CU cover each other's waits.
#ifndef CLITE_STEM_WGRAD_NW
#define CLITE_STEM_WGRAD_NW 4
#endif
template <int NW>
__global__ __launch_bounds__(NW * 64, 1) void stem_wgrad_patch_kernel(StemWgradArgs a, int strips_per_wg) {
  constexpr int NSTG = NW == 8 ? 2 : 1;
  constexpr int KSTEP = NW == 8 ? 2 : 1;
  __shared__ __attribute__((aligned(1024))) char smem[NSTG * SW_STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = wave_uniform(tid >> 6);
  const int kh = NW == 8 ? wave >> 2 : 0, rp = wave & 3;
  const int npix = 2 * a.Wo;                          // pixels of a full strip
  const int nks = npix >> 4;                          // k-steps per strip (Wo % 16 == 0: a k-step never straddles the two rows)
  const int ks_row = a.Wo >> 4;
  const int ndy = npix >> 3;                          // DMA instructions of the dy strip (8 pixels each)
  const int xbytes_strip = 9 * a.Wp * 8;
  const int nx = (xbytes_strip + 1023) >> 10;

  int s_begin = blockIdx.x * strips_per_wg, s_end = s_begin + strips_per_wg;
  if (s_end > a.nstrips) s_end = a.nstrips;
  const rsrc_t rdy = make_rsrc(a.dy, a.dybytes), rx = make_rsrc(a.x, a.xbytes);

  auto issue = [&](int strip, char* dst) {
    const int n = strip / a.strips_per_img;
    const int oy0 = (strip - n * a.strips_per_img) * 2;
    const int rows = a.Ho - oy0 < 2 ? a.Ho - oy0 : 2;          // (an odd Ho leaves a one-row strip: its second row gathers as zeros)
    const uint32_t dbase = (uint32_t)((n * a.Ho + oy0) * a.Wo) * PIXB;
    for (int i = wave; i < ndy; i += NW) {
      const int pix = i * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ (((pix >> 1) & 1) << 2);
      const bool v = pix < rows * a.Wo;
      PATCH_LOAD(rdy, v ? dbase + (uint32_t)(pix * PIXB + chunk * 16) : OOB_OFF, dst + i * 1024);
    }
    // the patch: 9 input rows from row 2 oy0, as they lie (rows past the image's padded height read as zeros: the buffer's bound, and for an image
    // that is not the last the one-row strip's second output row is masked by its zero dy)
    const uint32_t xbase = (uint32_t)((n * a.Hp + 2 * oy0) * a.Wp) * 8u;
    const uint32_t xend = (uint32_t)((n + 1) * a.Hp * a.Wp) * 8u;
    for (int i = wave; i < nx; i += NW) {
      const uint32_t off = xbase + (uint32_t)(i * 1024 + lane * 16);
      PATCH_LOAD(rx, (i * 1024 + lane * 16 < xbytes_strip && off < xend) ? off : OOB_OFF, dst + SW_DYB + i * 1024);
    }
  };

  // per-lane constants: A fragments (both channel halves) and the B fragment's offset inside a tap row's [pixel][32] image
  const int oa0 = wg_frag_base(0, 0, lane), oa1 = wg_frag_base(0, 32, lane);
  const int bl = 16 * (8 * (lane >> 5) + ((lane >> 2) & 3)) + 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
  const int r0 = 2 * rp;
  const bool two = rp < 3;
  f32x16 acc[2][2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][0][r] = 0.f; acc[0][1][r] = 0.f; acc[1][0][r] = 0.f; acc[1][1][r] = 0.f; }

  if (NSTG == 2 && s_begin < s_end) issue(s_begin, smem);
  for (int strip = s_begin; strip < s_end; ++strip) {
    const int cur = NSTG == 2 ? (strip - s_begin) & 1 : 0;
    if (NSTG == 1) {
      if (strip > s_begin) lds_barrier();          // every wave is past the K loop that read the stage
      issue(strip, smem);
    }
    wait_vmcnt<0>();
    lds_barrier();          // the strip's images landed (every wave's part); (two stages:) every wave is past the K loop that read the other stage
    if (NSTG == 2 && strip + 1 < s_end) issue(strip + 1, smem + (cur ^ 1) * SW_STAGE);
    const char* dyi = smem + cur * SW_STAGE;
    const char* xi = dyi + SW_DYB;
    // every wave multiplies TWO tap rows: for rp = 3 the second one is the non-existent row 7 - its reads stay inside the x region of the stage (patch
    // row 2 oyl + 7 <= 9: the launch bounds 10 Wp x 8 B by SW_XB) and its sums are never stored; a wave-uniform branch around it doubled the kernel's
    // accumulator registers (one set per instantiation: 266 registers, one wave per SIMD, 133 us instead of 60)
#pragma unroll 2
    for (int ks = kh; ks < nks; ks += KSTEP) {
      const int oyl = ks >= ks_row ? 1 : 0;
      const int ox0 = (ks - oyl * ks_row) << 4;
      const char* pa = dyi + ks * 16 * PIXB;
      const bf16x8 fa0 = wg_frag_at(pa + oa0), fa1 = wg_frag_at(pa + oa1);
      const char* pb = xi + ((2 * oyl + r0) * a.Wp + 2 * ox0) * 8 + bl;
      union { s16x4 v[2]; bf16x8 h; } b0, b1;
      b0.v[0] = lds_read_tr16(pb); b0.v[1] = lds_read_tr16(pb + 64);          // + 4 pixels
      b1.v[0] = lds_read_tr16(pb + a.Wp * 8); b1.v[1] = lds_read_tr16(pb + a.Wp * 8 + 64);
      acc[0][0] = mfma32_bf16(fa0, b0.h, acc[0][0]);
      acc[1][0] = mfma32_bf16(fa1, b0.h, acc[1][0]);
      acc[0][1] = mfma32_bf16(fa0, b1.h, acc[0][1]);
      acc[1][1] = mfma32_bf16(fa1, b1.h, acc[1][1]);
    }
  }

  // (8 waves) the odd-k-step waves hand their partial sums to the even ones through LDS ([c][224] f32), which add their own and write the slab
  auto at = [&](int cb, int ri, int r) { return (cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 224 + (r0 + ri) * 32 + (lane & 31); };
  if constexpr (NW == 4) {
    float* slab = a.ws + (size_t)blockIdx.x * SW_OUT;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int ri = 0; ri < 2; ++ri)
        if (ri == 0 || two)
#pragma unroll
          for (int r = 0; r < 16; ++r) slab[at(cb, ri, r)] = acc[cb][ri][r];
    return;
  }
  lds_barrier();
  float* fold = (float*)smem;
  if (kh == 1) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int ri = 0; ri < 2; ++ri)
        if (ri == 0 || two)
#pragma unroll
          for (int r = 0; r < 16; ++r) fold[at(cb, ri, r)] = acc[cb][ri][r];
  }
  lds_barrier();
  if (kh == 0) {
    float* slab = a.ws + (size_t)blockIdx.x * SW_OUT;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int ri = 0; ri < 2; ++ri)
        if (ri == 0 || two)
#pragma unroll
          for (int r = 0; r < 16; ++r) slab[at(cb, ri, r)] = acc[cb][ri][r] + fold[at(cb, ri, r)];
  }
}

// dw[c][r][s][ch] (f32 [64][7][7][3]) += sum over slabs of ws[b][c][r * 32 + s * 4 + ch]; grid (37, 8): a slice of the slabs per workgroup, one float
// atomic per output and slice
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nslabs) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 64 * 147) return;
  const int ch = i % 3, sx = (i / 3) % 7, r = (i / 21) % 7, c = i / 147;
  const int src = c * 224 + r * 32 + sx * 4 + ch;
  const int per = (nslabs + gridDim.y - 1) / gridDim.y;
  int b0 = blockIdx.y * per, b1 = b0 + per;
  if (b1 > nslabs) b1 = nslabs;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = b0;
  for (; b + 3 < b1; b += 4) {
    s0 += ws[(size_t)b * SW_OUT + src]; s1 += ws[(size_t)(b + 1) * SW_OUT + src]; s2 += ws[(size_t)(b + 2) * SW_OUT + src]; s3 += ws[(size_t)(b + 3) * SW_OUT + src];
  }
  for (; b < b1; ++b) s0 += ws[(size_t)b * SW_OUT + src];
  if (b0 < b1) atomic_add_f32(dw + i, (s0 + s1) + (s2 + s3));
}

}  // namespace

int clite::launch_stem_wgrad_patch(const void* dy, const void* xpad, int N, int Hp, int Wp, int Ho, int Wo, float* dw, void* ws, size_t ws_bytes, hipStream_t st) {
#ifdef CLITE_NO_PATCH
  return WIDE_NOT_TAKEN;
#endif
  if (Wo % 16 || 2 * Wo > SW_MAXPIX || 10 * Wp * 8 + 128 > SW_XB || Hp < 2 * (Ho - 1) + 7 || Wp < 2 * (Wo - 1) + 8) return WIDE_NOT_TAKEN;
  constexpr int WGS = (CLITE_STEM_WGRAD_NW == 8 ? 1 : 3) * CLITE_PATCH_WGS;
  if (!ws || ws_bytes < (size_t)WGS * SW_OUT * sizeof(float)) return WIDE_NOT_TAKEN;
  if ((size_t)N * Ho * Wo * PIXB >= 0xF0000000ull || (size_t)N * Hp * Wp * 8 >= 0xF0000000ull) return WIDE_NOT_TAKEN;
  StemWgradArgs a;
  a.dy = dy; a.x = xpad;
  a.dybytes = (uint32_t)((size_t)N * Ho * Wo * PIXB); a.xbytes = (uint32_t)((size_t)N * Hp * Wp * 8);
  a.N = N; a.Hp = Hp; a.Wp = Wp; a.Ho = Ho; a.Wo = Wo;
  a.strips_per_img = (Ho + 1) / 2;
  a.nstrips = N * a.strips_per_img;
  a.ws = (float*)ws;
  const int grid = a.nstrips < WGS ? a.nstrips : WGS;
  const int per = (a.nstrips + grid - 1) / grid;
  const int g2 = (a.nstrips + per - 1) / per;
  hipLaunchKernelGGL((stem_wgrad_patch_kernel<CLITE_STEM_WGRAD_NW>), dim3(g2), dim3(CLITE_STEM_WGRAD_NW * 64), 0, st, a, per);
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((64 * 147 + 255) / 256, 8), dim3(256), 0, st, (const float*)ws, dw, g2);
  return (int)hipGetLastError();
}

size_t clite::conv3x3_wgrad_patch_workspace() {          // the larger of its two users' needs (the stem's weight gradient runs three workgroups per CU)
  const size_t a = (size_t)CLITE_PATCH_WGS * WG_OUT * sizeof(float), b = (size_t)3 * CLITE_PATCH_WGS * 64 * 224 * sizeof(float);
  return a > b ? a : b;
}

int clite::launch_conv3x3_wgrad_patch(const void* dy, const void* x, const clite_conv& c, float* dw, void* ws, size_t ws_bytes, hipStream_t st) {
#ifdef CLITE_NO_PATCH
  return WIDE_NOT_TAKEN;
#endif
  if (c.dtype != CLITE_BF16 || c.C != PC || c.K != PC || c.R != 3 || c.S != 3 || c.stride != 1 || c.pad != 1 || c.Ho != c.H || c.Wo != c.W) return WIDE_NOT_TAKEN;
  if (!ws || ws_bytes < conv3x3_wgrad_patch_workspace() || c.W + 1 > 59) return WIDE_NOT_TAKEN;
  int R = WG_DYPIX / (c.W + 1);
  if (R > c.H) R = c.H;
  while (R >= 1 && ((R + 2) * (c.W + 1) + 1 > WG_XPIX || WG_DYPIX + 2 * (c.W + 1) + 2 > WG_XPIX)) --R;
  if (R < 1) return WIDE_NOT_TAKEN;
  WgradPatchArgs a;
  a.dy = dy; a.x = x; a.bytes = (uint32_t)((size_t)c.N * c.H * c.W * PC * 2);
  a.N = c.N; a.H = c.H; a.W = c.W; a.R = R;
  a.strips_per_img = (c.H + R - 1) / R;
  a.nstrips = c.N * a.strips_per_img;
  a.ws = (float*)ws;
  const int grid = a.nstrips < CLITE_PATCH_WGS ? a.nstrips : CLITE_PATCH_WGS;
  const int per = (a.nstrips + grid - 1) / grid;
  const int g2 = (a.nstrips + per - 1) / per;
  hipLaunchKernelGGL(conv3x3_wgrad_patch_kernel, dim3(g2), dim3(512), 0, st, a, per);
  hipLaunchKernelGGL(wgrad_patch_reduce_kernel, dim3(WG_OUT / 256, 16), dim3(256), 0, st, (const float*)ws, dw, g2);
  return (int)hipGetLastError();
}
