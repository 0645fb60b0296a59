/* clite.h — C ABI of libclite_hip.so, the MI355X (gfx950) kernel library behind the CLIP-Lite
 * pretraining step.
 *
 * The reference (4m4n5/CLIP-Lite) has no native layer: its hot path is Python calling ATen/cuDNN/cuBLAS.
 * Every entry point below therefore cites the reference *call site* whose device math it replaces
 * (file:line relative to the reference repository). The host side that mirrors the reference's Python
 * classes lives in clip-lite_amd/ and reaches these functions through ctypes.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are device pointers owned by the caller
 *   - bf16 tensors are passed as void*; "f32" means IEEE binary32
 *   - activations are NHWC / row-major [rows][features]; weights are [out][...in], inner dim contiguous
 *   - `stream` is a hipStream_t; kernels are only enqueued, nothing synchronises, nothing allocates
 *   - return value: 0 on success, a hipError_t value or -1 (bad argument) otherwise; nothing throws
 *   - entry points are re-entrant (forward runs on the Python main thread, backward on the autograd engine
 *     thread). The library holds exactly two pieces of process-wide state, both plain atomics set through their
 *     own entry points and never written by a kernel launcher: the deterministic-reduction flag
 *     (clite_set_deterministic) and the tile policy (clite_set_tile_policy). Both are read once per call, when
 *     the kernels are ENQUEUED: a captured hipGraph keeps the mode it was recorded in, an eager step follows the
 *     current value — set them before a step, not during one.
 */
#ifndef CLITE_H
#define CLITE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLITE_ABI_VERSION 12
int clite_abi_version(void);

/* Deterministic-reduction mode (process-wide, default off; the counterpart of torch.use_deterministic_algorithms for this library).
 * Off: split-K partial products, BatchNorm / bias / LayerNorm statistics and the loss scalars meet through float atomics, so the
 * summation order — and with it the last bits of every gradient — differs from run to run. On: every launcher picks a decomposition in
 * which each address receives exactly one contribution per launch (unsplit K ranges, one row slab per statistics replica, single-wave
 * loss kernels; clip-lite_amd/csrc/det.h), so two runs of the same step are bit-identical — which is what makes checkpoint-resume
 * equivalence (reference utils/checkpointing.py:169-222, train.py:143-148) testable exactly. Several times slower; a testing aid.
 * The flag is read when a kernel is enqueued (a captured hipGraph keeps the mode it was recorded in). Returns 0. */
int clite_set_deterministic(int on);
int clite_get_deterministic(void);

/* ABI v10 — split-bf16 form of the exact-f32 mode (process-wide, like the deterministic switch). With dtype CLITE_F32 the GEMM / convolution entry
 * points keep f32 storage, f32 accumulation and f32 epilogues, but form every product on three bf16 MFMAs: x = hi + lo with hi = bf16(x),
 * lo = bf16(x - hi), a b ~ lo_a hi_b + hi_a lo_b + hi_a hi_b (bf16 x bf16 products are exact in f32). Relative error ~2^-17 per product — between the
 * bf16 path's 2^-9 and the default f32 path's exact fmaf chain (v_mfma_f32_32x32x2_f32, 1/16 of the bf16 matrix rate) — at 3/16 of the default's
 * matrix-pipe time. The full-size parity test (tests/test_gpu_model.py) holds this form to the same 1e-4 loss bar as the exact form. Off by default;
 * bf16 launches are unaffected. */
int clite_set_f32_split(int on);
int clite_get_f32_split(void);

/* Tile-shape policy of the bf16 GEMM / conv launchers (process-wide, default 0). 0: automatic — per launch, the largest of the 128 x 128,
 * 256 x 128 and 256 x 256 wide-K tiles (K tile 64 = whole 128-byte lines, 8 waves; clip-lite_amd/csrc/igemm_wide.h) that still fills the
 * 256 CUs. 1 / 2 / 3: force that wide tile wherever an instantiation exists. 4: keep every launch on the 4-wave 128 x 128 x 32 kernels.
 * The forced forms exist so that every instantiation can be parity-tested on small problems; results are identical up to the
 * summation order of the K loop. Returns 0, or -1 for an unknown policy. */
int clite_set_tile_policy(int policy);

/* storage type of activations / weights handed to a kernel. CLITE_F32 selects the exact-f32 parity mode
 * (v_mfma_f32_32x32x2_f32, 1/16 of the bf16 MFMA rate). Accumulation is f32 in both. */
enum { CLITE_BF16 = 0, CLITE_F32 = 1 };

/* Dropout / noise seeds. Every (seed, site) pair below is normally passed by value. When CLITE_SEED_INDIRECT is OR-ed into the
 * site, the 64-bit seed argument is instead the DEVICE ADDRESS of a uint64_t holding the seed: the kernel loads it (one scalar
 * load per wave) and masks the flag off the site. A captured hipGraph of the train step uses this form so that every replay
 * draws fresh masks from a device-side counter without re-recording kernel arguments. */
#define CLITE_SEED_INDIRECT 0x80000000u

enum { CLITE_ACT_NONE = 0, CLITE_ACT_RELU = 1, CLITE_ACT_GELU = 2, CLITE_ACT_TANH = 3 };

/* Fused GEMM epilogue. v = alpha*acc + bias; preact <- v; v = act(v); v *= act'(dact_aux);
 * v = dropout(v); v += residual; [mask_after_residual: the act'(dact_aux) factor is applied here instead]; out <- v;
 * colsum += (sum, sum of squares) of the stored values — or, when bn_y is given, (sum v, sum v*(bn_y - mean)): the two
 * reductions of a BatchNorm backward over the tensor this GEMM produces (dz = masked gradient, bn_y = that BN's input,
 * mean_c = sum over replicas of bn_stats[r*bn_rstride + c] * bn_inv_count), which saves the separate reduction pass. */
typedef struct clite_epilogue {
  void* out;            /* [M][ldc] in the call's dtype, or f32 when out_f32 = 1 */
  int32_t ldc;
  int32_t out_f32;
  int32_t atomic;       /* 1: out (f32) += alpha*acc with float atomics (split-K / gradient accumulation) */
  float alpha;
  const float* bias;    /* [N] f32 or NULL */
  int32_t act;          /* CLITE_ACT_* */
  void* preact;         /* [M][ldc] in the call's dtype, or NULL */
  const void* dact_aux; /* [M][ldc] in the call's dtype, or NULL */
  int32_t dact;         /* 1 relu' (aux = forward output), 2 gelu' (aux = pre-activation), 3 tanh' (aux = output) */
  float drop_p;
  uint64_t drop_seed;
  uint32_t drop_site;
  const void* residual; /* [M][ldc] in the call's dtype, or NULL */
  float* colsum;        /* f32 [2][N] or NULL */
  int32_t colsum_replicas;  /* R > 1: workgroup b accumulates into colsum + (b % R) * colsum_stride (spreads same-address atomics) */
  int32_t colsum_stride;    /* elements between replicas */
  int32_t colsum_rows;      /* 0 or 2: both rows [2][N]; 1: the column sums only (row 0) — e.g. straight into a bias gradient, whose neighbour in
                             * memory is another tensor */
  const void* bn_y;         /* [M][ldc] in the call's dtype, or NULL (plain sum of squares) */
  const float* bn_stats;    /* forward statistics of that BatchNorm: replicated [R][3][N] sums (row 0 = sum of y) */
  int32_t bn_replicas;
  int32_t bn_rstride;
  float bn_inv_count;       /* 1 / rows of the BatchNorm */
  int32_t mask_after_residual;
  const uint8_t* relu_bits; /* BatchNorm-backward form only (bn_y / mask_after_residual): the relu' mask as PACKED BITS instead of dact_aux — bit (c & 7)
                             * of byte (row * ldc + c) / 8 is 1 where the forward activation was > 0 (written by clite_bn_apply, clite_bn.relu_bits):
                             * 1/16 of the bytes of reading the bf16 activation for its sign. NULL: dact_aux (if any) supplies the mask. */
  float* splitk_ws;         /* optional f32 [M][N] workspace, ZERO on entry (left dirty): lets clite_gemm_nt / _nn run a GEMM of few output tiles
                             * (the M = 128/256-row GEMMs of the projection heads and prior discriminators) as split-K over all CUs — partial
                             * sums accumulate here with float atomics and a second small kernel applies this epilogue. NULL: never split. */
  int32_t residual_subsample; /* ABI v10. 0 / 1: `residual` has the output's rows. 2: clite_conv_dgrad_wt of a 1 x 1 / stride 1 convolution in the
                             * BatchNorm-backward form only (bn_y + relu_bits + mask_after_residual, bf16): `residual` is the COMPACT tensor
                             * [N][H/2][W/2][ldc] whose pixel (n, h/2, w/2) is added to output pixel (n, h, w) for even h, w and nothing elsewhere —
                             * the input gradient of the block's stride-2 1 x 1 shortcut, kept compact instead of scatter-added into the full-size
                             * gradient (which cost a read-modify-write pass and, before it, a separate BatchNorm-backward reduction pass). */
  /* ABI v11, clite_gemm_nt_fp8 only (every other entry point returns -1 when one of them is set): the producer-fused e4m3 quantiser of the NEXT
   * GEMM's A operand, as clite_bn.fp8_* for clite_bn_apply — BERT's FFN1 launch leaves the e4m3 copy of its GELU output for FFN2.
   *   fp8_out   [M][ldc] bytes: e4m3(clamp(out * fp8_scale[0], +-448)) of the bf16 value as stored (bf16 out only);
   *   fp8_scale device f32[2] = {scale, 1 / scale}, delayed scaling (clite_fp8_scale_update); required with fp8_out;
   *   fp8_amax  one amax slot (CLITE_FP8_AMAX_REPLICAS x CLITE_FP8_AMAX_STRIDE words): max |out| of this call. Any of the three may be NULL. */
  uint8_t* fp8_out;
  const float* fp8_scale;
  float* fp8_amax;
} clite_epilogue;

/* NHWC convolution problem. x: [N][H][W][C], w: [K][R][S][C], y: [N][Ho][Wo][K], all of `dtype`. */
typedef struct clite_conv {
  int32_t dtype;
  int32_t N, H, W, C;
  int32_t K, R, S;
  int32_t stride, pad;
  int32_t Ho, Wo;
} clite_conv;

/* lda / ldb: row strides in elements (multiples of 8), so a strided row subset (BERT's h[:, 0] for the pooler) needs no copy.
 * C[M,N] = A[M,K] * B[N,K]^T  — nn.Linear forward (reference loss.py:16-22,46-48; HF BertModel linears
 * behind encoder.py:193). K % 8 == 0, N % 8 == 0. */
int clite_gemm_nt(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int dtype, const clite_epilogue* ep, void* stream);
/* C[M,N] = A[M,K] * B[K,N]       — nn.Linear input gradient (autograd of the same call sites). */
int clite_gemm_nn(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int dtype, const clite_epilogue* ep, void* stream);
/* C[M,N] (+)= A[K,M]^T * B[K,N]  — nn.Linear weight gradient; ep->atomic selects f32 accumulation. */
int clite_gemm_tn(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int dtype, const clite_epilogue* ep, void* stream);

/* y = conv(x, w): torchvision ResNet nn.Conv2d forward (reference encoder.py:36-38,63; block arithmetic as
 * restated in model_zoo/resnet.py:60-100). Epilogue applies to y viewed as [N*Ho*Wo][K]. */
int clite_conv_fwd(const void* x, const void* w, const clite_conv* cv, const clite_epilogue* ep, void* stream);
/* dx = conv_transpose(dy, w): autograd of the same call. Epilogue applies to dx viewed as [N*H*W][C]. */
int clite_conv_dgrad(const void* dy, const void* w, const clite_conv* cv, const clite_epilogue* ep, void* stream);
/* One parity class (ph, pw in {0,1}) of the dgrad of a 3x3 / stride-2 / pad-1 conv (H, W even; C, K multiples of 64): writes only the
 * input pixels (2*hq + ph, 2*wq + pw). wsub = the taps that reach this class, w[:, r0::2, s0::2, :] with r0 = (ph+1)&1, s0 = (pw+1)&1,
 * packed [K][na][nb][C]. The four classes together equal clite_conv_dgrad at a quarter of the work (no structurally-zero taps). The
 * epilogue (incl. the BatchNorm-backward form) applies to the rows of the class. */
int clite_conv_dgrad_s2class(const void* dy, const void* wsub, const clite_conv* cv, int ph, int pw, const clite_epilogue* ep, void* stream);
/* The same two input gradients with the weight given TRANSPOSED, wt = [C][R][S][K] (wtsub = [C][na][nb][K]): both GEMM operands are then
 * k-contiguous — the forward form of the tile engine — which the 8-wave kernels stage 10-25 % faster than a k-strided weight image. The
 * transposed copies are re-derived once per step by clite_transpose_weights. Same epilogue semantics and results (up to the summation order). */
int clite_conv_dgrad_wt(const void* dy, const void* wt, const clite_conv* cv, const clite_epilogue* ep, void* stream);
int clite_conv_dgrad_s2class_wt(const void* dy, const void* wtsub, const clite_conv* cv, int ph, int pw, const clite_epilogue* ep, void* stream);
/* dw[K][R][S][C] (f32) += dy^T * im2col(x): autograd of the same call; float-atomic split-K. */
int clite_conv_wgrad(const void* dy, const void* x, const clite_conv* cv, float* dw, void* stream);
/* ABI v10 — the same weight gradient for the 3 x 3 / stride 1 / pad 1, 64 -> 64 channel convolutions (torchvision ResNet layer1 conv2; bf16) on the
 * patch-resident kernel: one persistent workgroup per CU owns the whole [64][3][3][64] gradient in registers and reads dy and the input patch of
 * each strip of rows ONCE (the grouped form gathers every input pixel nine times: 97 us against an HBM ideal of 19 at 56 x 56, batch 128).
 * ws: caller-owned scratch of at least clite_conv_wgrad_patch_workspace() bytes (per-workgroup partial sums; contents undefined afterwards).
 * Returns 0 when launched, 1 when the problem is not one it covers (other shape / dtype, workspace too small, deterministic mode, forced tile
 * policy): nothing was launched and the caller takes clite_conv_wgrad or clite_wgrad_group; < 0 on errors. */
int clite_conv_wgrad_patch_workspace(uint64_t* nbytes);
int clite_conv_wgrad_patch(const void* dy, const void* x, const clite_conv* cv, float* dw, void* ws, uint64_t ws_bytes, void* stream);

/* Grouped weight gradients: every member is one independent dW (f32) += A^T B of the backward pass, all enqueued as ONE launch per tile
 * family so that thousands of workgroups exist without splitting the short-K members (a lone weight gradient has a few dozen output tiles
 * and needs split-K + float atomics to fill 256 CUs: 328 vs 666 TF/s on the BERT FFN gradient; clip-lite_amd/csrc/gemm_group.hip).
 *   kind 0: conv weight gradient — a = dy [N][Ho][Wo][K], b = x [N][H][W][C], cv, out = dw [K][R][S][C]   (as clite_conv_wgrad)
 *   kind 1: linear weight gradient — out[M][N] (row stride ldc) += a[K][M]^T b[K][N]                     (as clite_gemm_tn with ep.atomic)
 *   kind 2 (ABI v12; BASELINE configs[4]): conv weight gradient on FP8 operands — a = dy8 [N][Ho][Wo][K] OCP e5m2 (the copy clite_bn_bwd_apply's fused
 *           quantiser leaves for clite_conv_dgrad_fp8), b = x8 [N][H][W][C] OCP e4m3 (the copy clite_bn_apply leaves for clite_conv_fwd_fp8), cv, a_scales /
 *           b_scales; C % 16 == 0, K % 16 == 0. v_mfma_scale_f32_32x32x64_f8f6f4 (A e5m2, B e4m3) at unit block scales on the grouped 256 x 256 tile,
 *           fragments by ds_read_b64_tr_b8, f32 accumulate. Grouped launches only (-1 where the members run one by one).
 * ws_dev / ws_host: device workspace and PINNED host staging of ws_bytes each (clite_wgrad_group_workspace gives a sufficient size); the
 * library fills ws_host, copies it with one hipMemcpyAsync on `stream` and launches — under stream capture that is a memcpy node, so ws_host
 * must stay alive and unchanged for as long as the captured graph is replayed. With dtype = CLITE_F32, in deterministic-reduction mode, or
 * with ws_dev / ws_host NULL the members are launched one by one. Returns 0, -1 (bad member), -2 (workspace too small) or a HIP status. */
#define CLITE_WGRAD_NARROW 0x100   /* OR-ed into `kind`: keep this member on the 4-wave 128 x 128 x 32 tiles (members with >= 256 rows and columns
                                    * otherwise take the 8-wave 256 x 256 tile); for A/B timing and parity tests */
#define CLITE_WGRAD_ZEROED 0x200   /* ABI v10, OR-ed into `kind`: the caller vouches that `out` holds ZEROS on entry (the train step's update kernel leaves the
                                    * gradient arena zeroed and the step visits this weight once). A member whose contraction fits one K chunk — every BERT
                                    * matrix, the 7 x 7-resolution convolutions — then writes its result with plain stores instead of float atomics
                                    * (memory-side, ~1.3 TB/s chip-wide: 437 MB per step for BERT's 109 M parameters). Same values; without the flag `+=`. */
#define CLITE_WGRAD_SHORTK 0x400   /* ABI v12, OR-ed into `kind`: K chunks of a quarter of the usual length — a member with a tiny output and a very long contraction (the
                                    * folded BatchNorm backward's Gram matrix a^T a: 64 x 64 ... 128 x 128 outputs over 100k - 400k pixels), whose few long workgroups
                                    * would otherwise be the tail of the launch; more chunks cost only more float atomics into that tiny output */
typedef struct clite_wgrad_item {
  int32_t kind;
  const void* a;
  const void* b;
  float* out;
  clite_conv cv;                 /* kind 0 */
  int32_t M, N, K, lda, ldb, ldc;  /* kind 1 */
  const float* a_scales;         /* ABI v12, kind 2 only: device f32 {scale, 1 / scale} of a (clite_bn.fp8_scale of the pass that wrote it) ... */
  const float* b_scales;         /* ... and of b; out += a_scales[1] * b_scales[1] * (a^T b) */
  const float* row_scale;        /* ABI v12. NULL, or device f32 [rows of out]: out[r][:] += row_scale[r] * (A^T B)[r][:] — the folded BatchNorm backward's
                                  * per-channel factor ka (clite_bn_fold_prepare) on a weight gradient contracted against dz instead of dy. Grouped
                                  * launches only: -1 where the members run one by one (f32, deterministic mode, no workspace). */
} clite_wgrad_item;
int clite_wgrad_group(int dtype, const clite_wgrad_item* items, int n, void* ws_dev, void* ws_host, uint64_t ws_bytes, void* stream);
int clite_wgrad_group_workspace(int n_items, int64_t total_workgroups, uint64_t* bytes);

/* Batched bf16 matrix transposes inside two flat buffers (the bf16 weight arena and its transposed twin): item i copies `batch` matrices
 * src[src_off + b*src_bstride + r*src_ld + c] -> dst[dst_off + b*dst_bstride + c*dst_ld + r], r < rows, c < cols (rows, cols, the strides
 * and offsets multiples of 8 elements). One launch for all items: workgroup w handles one 64 x 64 tile of the item with first_tile <= w <
 * first_tile + batch * ceil(rows/64) * ceil(cols/64); `items_dev` is a DEVICE array sorted by first_tile, total_tiles the sum. Linear weight
 * [N][K] -> [K][N]: rows N, cols K, batch 1. Conv weight [K][R][S][C] -> [C][R][S][K]: rows K, cols C, src_ld R*S*C, dst_ld R*S*K, batch R*S,
 * src_bstride C, dst_bstride K. (Reference: nothing — autograd's input gradients read the same weight tensor; this is a layout choice.) */
typedef struct clite_transpose_item {
  uint64_t src_off, dst_off;
  uint32_t rows, cols, src_ld, dst_ld, batch, src_bstride, dst_bstride, first_tile;
} clite_transpose_item;
int clite_transpose_weights(const void* src, void* dst, const clite_transpose_item* items_dev, int n_items, uint32_t total_tiles, void* stream);

/* ---- OCP e4m3 forward path (BASELINE.json configs[4]; the reference has no fp8 code — the policy is this library's).
 * Per-tensor current scaling: amax = max|x| (ZERO on entry), scale = 448 / amax, q = e4m3(clamp(x * scale, +-448)) rounded to nearest even;
 * scales (device f32[2]) <- {scale, 1 / scale}. n % 8 == 0. dtype of x: CLITE_BF16 or CLITE_F32. */
int clite_fp8_quantize(int dtype, const void* x, uint64_t n, float* amax, float* scales, void* out_fp8, void* stream);
/* C[M,N] = (A8[M,K] * B8[N,K]^T) * a_scales[1] * b_scales[1] through the fused epilogue (bf16 / f32 output; no split-K, no BatchNorm-backward
 * form): nn.Linear forward on e4m3 operands, v_mfma_f32_32x32x16_fp8_fp8 with f32 accumulation. K, lda, ldb multiples of 16. */
int clite_gemm_nt_fp8(const void* A8, int lda, const void* B8, int ldb, int M, int N, int K, const float* a_scales, const float* b_scales,
                      const clite_epilogue* ep, void* stream);
/* y = conv(x8, w8) * x_scales[1] * w_scales[1]: clite_conv_fwd on e4m3 operands (x8 [N][H][W][C], w8 [K][R][S][C]); C % 16 == 0, and
 * C % 64 == 0 for windowed convs. */
int clite_conv_fwd_fp8(const void* x8, const void* w8, const clite_conv* cv, const float* x_scales, const float* w_scales,
                       const clite_epilogue* ep, void* stream);
/* ABI v11. Input gradient of a stride-1 convolution on fp8 operands, in the one form the bf16 ResNet backward issues for a unit inside a block:
 * dx [N][H][W][C] bf16 = relu'(ep->relu_bits) * (dy (*) w), with the two BatchNorm-backward reductions (ep->bn_y, ep->bn_stats, ep->colsum) - every
 * other clite_epilogue feature is refused. dy8: [N][Ho][Wo][K] OCP e5m2 (written by clite_bn_bwd_apply's producer-fused quantiser: clite_bn.fp8_*
 * of that call; gradients get e5m2's range), wt8: the TRANSPOSED weights [C][R][S][K] in OCP e4m3; dy_scales / w_scales: device f32 {scale, 1 / scale}.
 * K % 64 == 0, C % 8 == 0. v_mfma_scale_f32_32x32x64_f8f6f4 (A e5m2, B e4m3) at unit block scales, f32 accumulate. */
int clite_conv_dgrad_fp8(const void* dy8, const void* wt8, const clite_conv* cv, const float* dy_scales, const float* w_scales,
                         const clite_epilogue* ep, void* stream);
/* Delayed scaling, once per step over all producer-fused tensors: slot i of `amax` (clite_bn.fp8_amax; a = the maximum of its words) with
 * a != 0: scales[2i] = 448 / a, scales[2i + 1] = a / 448 (both NaN when a is not finite); the slot is zeroed. A slot that stayed 0 keeps
 * its scales. */
#define CLITE_FP8_AMAX_REPLICAS 16
#define CLITE_FP8_AMAX_STRIDE 32
int clite_fp8_scale_update(float* amax, float* scales, int n, void* stream);
/* Per-tensor CURRENT scaling of many tensors of one arena in two launches (the conv weights after the optimizer update): item i is the
 * `numel` (% 8 == 0) elements at element offset `offset` (% 8 == 0) of `base`; its e4m3 copy goes to q_base + offset, its scales to
 * scales[2i..2i+1], its amax to amax[i]. wg_table_dev[w] = (item << 12 | chunk): workgroup w handles elements [chunk * 8192, (chunk + 1) * 8192)
 * of its item (chunk < 4096, i.e. tensors up to 32 M elements; items < 2^20), an item's chunks CONSECUTIVE and ascending in the table;
 * `partial` is n_wgs words of scratch. No atomics, nothing to zero: the result is a pure function of the arena. bf16 only. */
typedef struct clite_fp8_item {
  uint64_t offset, numel;
} clite_fp8_item;
int clite_fp8_quantize_group(const void* base, const clite_fp8_item* items_dev, const uint32_t* wg_table_dev, int n_items, int n_wgs,
                             uint32_t* partial, float* amax, float* scales, void* q_base, void* stream);

/* ResNet stem conv1 = nn.Conv2d(3, 64, 7, stride 2, padding 3, bias=False) (torchvision, reference encoder.py:36-38) on the
 * pre-padded NHWC4 image from clite_image_to_nhwc4 (Hp >= 2*(Ho-1)+7, Wp >= 2*(Wo-1)+8, Wp even). wv is the weight packed
 * as [64][7][8][4] (clite_stem_pack from f32 [64][7][7][3]); clite_stem_unpack_grad folds the packed gradient back (+=). */
int clite_stem_fwd(const void* xpad, const void* wv, int dtype, int N, int Hp, int Wp, int Ho, int Wo, const clite_epilogue* ep, void* stream);
int clite_stem_wgrad(const void* dy, const void* xpad, int dtype, int N, int Hp, int Wp, int Ho, int Wo, float* dwv, void* stream);
/* ABI v11. The same weight gradient on a patch-resident kernel (bf16, Wo % 16 == 0, Wo <= 128, Wp <= 254): dw f32 [64][7][7][3] += directly (no
 * packed intermediate), through clite_conv_wgrad_patch_workspace() bytes of scratch. Returns 1 (nothing launched) when the problem is not covered -
 * the caller then takes clite_stem_wgrad + clite_stem_unpack_grad - or in the deterministic mode / under a forced tile policy. */
int clite_stem_wgrad_patch(const void* dy, const void* xpad, int dtype, int N, int Hp, int Wp, int Ho, int Wo, float* dw, void* ws, uint64_t ws_bytes,
                           void* stream);
int clite_stem_pack(const float* w, void* wv, int dtype, void* stream);
int clite_stem_unpack_grad(const float* dwv, float* dw, void* stream);

/* ---- BatchNorm2d / BatchNorm1d, train-mode batch statistics (torchvision ResNet BN behind reference
 * encoder.py:36-65; nn.BatchNorm1d at loss.py:18). Tensors are [M][C] (NHWC flattened), C % 8 == 0, C/8 divides 256.
 * `stats` = per-channel (sum, sum of squares) [2][C] f32 over the M rows, produced by the conv/GEMM epilogue
 * (clite_epilogue.colsum). Biased variance normalises; unbiased variance goes into running_var. */
typedef struct clite_bn {
  int32_t M, C;
  const float* stats;        /* [2][C], or [3][C] when centered = 1 */
  const float* gamma;        /* [C] */
  const float* beta;         /* [C] */
  float* running_mean;       /* [C] */
  float* running_var;        /* [C] */
  int32_t training;          /* 1: batch statistics; 0: running statistics (eval) */
  int32_t update_running;    /* 1: workgroup 0 updates running_* with `momentum` */
  float momentum, eps;
  int32_t relu;              /* apply ReLU at the end */
  int32_t replicas;          /* R >= 1: `stats` (and res_stats) are R partial accumulators `rstride` elements apart; readers sum them */
  int32_t rstride;
  int32_t centered;          /* 1: variance = stats[2][c] / M (two-pass, from clite_bn_centered_var) instead of E[x^2]-E[x]^2 */
  /* optional second BN applied to the residual operand (the 1x1/stride downsample branch of a block) */
  const float* res_stats;
  const float* res_gamma;
  const float* res_beta;
  float* res_running_mean;
  float* res_running_var;
  uint8_t* relu_bits;        /* clite_bn_apply with relu = 1: optional [M][C / 8] bytes, bit e of byte (m * C + c) / 8 = (out[m][c + e] > 0): the
                              * ReLU mask the backward pass needs, at 1/16 of the bytes of re-reading `out` for its sign. NULL: not written. */
  /* clite_bn_apply (e4m3) and, ABI v11, clite_bn_bwd_apply (the same three fields, OCP e5m2 codes of dy for clite_conv_dgrad_fp8: clamp at +-57344, the
   * scale still maps the recorded amax to 448, i.e. 128 x headroom for a gradient that grew since); bf16; any of the three may be NULL:
   *   fp8_out   [M][C] e4m3 copy of `out` for the convs that read it: q = e4m3(clamp(out * fp8_scale[0], +-448)) of the value as stored,
   *             written by the same pass — half a write instead of clite_fp8_quantize's two reads and half a write;
   *   fp8_scale device f32[2] = {scale, 1 / scale}: DELAYED scaling, made from an earlier step's amax (clite_fp8_scale_update);
   *   fp8_amax  one amax SLOT = device f32[CLITE_FP8_AMAX_REPLICAS * CLITE_FP8_AMAX_STRIDE]: max |out| of THIS call folded in by integer
   *             atomic maxima on the bit pattern, workgroup b into word (b % REPLICAS) * STRIDE — same-address atomics retire at ~90 per
   *             microsecond chip-wide, a thousand workgroups on one word cost more than the pass over a small tensor; the slot's value is
   *             the maximum of its words (clite_fp8_scale_update folds and zeroes them). A NaN anywhere makes it the quiet-NaN pattern,
   *             an infinity +inf: both give NaN scales, as clite_fp8_quantize does. */
  uint8_t* fp8_out;
  const float* fp8_scale;
  float* fp8_amax;
  /* ABI v12, clite_bn_apply only (bf16, without the fp8 fields): out_sum[(b % out_sum_replicas) * out_sum_stride + c] += sum over workgroup b's rows of
   * out[m][c] as stored — the column sums of the activation that the folded BatchNorm backward's weight gradient needs (clite_bn_fold_wgrad_finish),
   * taken by the pass that writes it instead of a pass of their own. ZERO on entry. NULL: not accumulated. */
  float* out_sum;
  int32_t out_sum_replicas;
  int32_t out_sum_stride;
} clite_bn;

/* Second pass of a two-pass variance (used by the exact-f32 parity mode): stats[2][c] += sum_m (y[m][c] - stats[0][c]/M)^2;
 * stats[2][*] must be zero on entry. */
int clite_bn_centered_var(int dtype, const void* y, float* stats, int replicas, int rstride, int M, int C, void* stream);
/* out = relu?( bn(y) + [res | bn_res(res)] ) */
int clite_bn_apply(const clite_bn* p, int dtype, const void* y, const void* res, void* out, void* stream);
/* dstats[0][c] += sum dz, dstats[1][c] += sum dz*(y - mean_c), dz = dout * (mask > 0) (mask NULL: dz = dout), mean_c = stats[0][c]/M.
 * The mask is either a tensor of the call's dtype (`mask`) or packed bits as clite_bn.relu_bits writes them (`mask_bits`); at most one of
 * the two. dstats pre-zeroed. */
int clite_bn_bwd_reduce(int dtype, const void* dout, const void* mask, const uint8_t* mask_bits, const void* y, const float* stats, float* dstats,
                        int replicas, int rstride, int M, int C, void* stream);   /* stats and dstats are both replicated (R, rstride) */
/* dy = BN backward of dz through batch statistics; dz (optional) <- masked dout; dgamma/dbeta (optional) += . */
int clite_bn_bwd_apply(const clite_bn* p, int dtype, const void* dout, const void* mask, const uint8_t* mask_bits, const void* y, const float* dstats,
                       void* dy, void* dz, float* dgamma, float* dbeta, void* stream);

/* ---- ABI v12: the block-output BatchNorm backward FOLDED into its consumers (bf16; reference model_zoo/resnet.py:60-100 block arithmetic, autograd of
 * encoder.py:63's torchvision Bottleneck: bn3 backward -> conv3 input gradient + conv3 weight gradient). BatchNorm backward is linear in dz:
 *     dy = ka . dz + kb + kc . (y - mean),   ka = gamma rstd,  kb = -ka S1 / M,  kc = -ka rstd^2 S2 / M     (per channel k; S1 = sum dz, S2 = sum dz (y - mean))
 * so for the 1 x 1 convolution y = a W^T (W [K][Cin]) that produced y
 *     da = dy W          = [dz | y] [diag(ka) W ; diag(kc) W] + (kb - kc . mean) W       one GEMM over the K-concatenation of dz and y, a bias row
 *     dW = dy^T a        = diag(ka) (dz^T a) + kb (x) colsum(a) + diag(kc) W Cov(a),     Cov(a) = a^T a - colsum(a) colsum(a)^T / M   (y = a W^T, mean = colsum(a) W^T / M)
 * and the apply pass (read dz, read y, write dy) with its dy tensor disappears: the dgrad reads dz and y instead of dy, the weight gradient reads dz instead of dy.
 *
 * clite_bn_fold_prepare: from the BatchNorm's forward sums `p->stats`, its two backward reductions `dstats` (same replica layout) and the TRANSPOSED bf16
 * weights wt [Cin][K] (K = p->C) makes, in one launch of Cin workgroups:
 *   w2   bf16 [Cin][2][K]   w2[c][0][k] = bf16(kc[k] wt[c][k]) (multiplies y), w2[c][1][k] = bf16(ka[k] wt[c][k]) (multiplies dz): clite_conv_dgrad_bnfold's weights
 *   bias f32  [Cin]         sum_k kb[k] wt[c][k] - sum_k mean[k] w2[c][0][k]   (with the ROUNDED w2: sum_k (y - mean) w2 then cancels exactly as in y - mean)
 *   coef f32  [3][K]        ka, kb, kc
 *   dgamma[k] += rstd S2, dbeta[k] += S1 (either may be NULL). */
int clite_bn_fold_prepare(const clite_bn* p, const float* dstats, const void* wt, int Cin, void* w2, float* bias, float* coef, float* dgamma, float* dbeta,
                          void* stream);
/* da [M][Cin] = [pair[0] | pair[1]] [w2[:, 1, :] | w2[:, 0, :]]^T + ep->bias through the BatchNorm-backward form of the epilogue (relu_bits + bn_y + colsum:
 * the NEXT BatchNorm backward's mask and reductions), i.e. what clite_bn_bwd_apply followed by clite_conv_dgrad_wt computes. pair = bf16 [2][M][K]: slot 0 the
 * masked gradient dz, slot 1 the BatchNorm input y (the producer of each writes it there). K % 64 == 0, Cin % 8 == 0, 2 M K < 2^31. */
int clite_conv_dgrad_bnfold(const void* pair, const void* w2, int M, int K, int Cin, const clite_epilogue* ep, void* stream);
/* Weight-gradient side: the two correction terms, in f32, with G = a^T a (f32 [Cin][Cin], left by a member of the grouped launch) and the column sums
 * s[c] = sum_r asum[r * asum_stride + c] (clite_bn.out_sum of the clite_bn_apply that wrote a):
 *   dw[k][c] += kb[k] s[c] + kc[k] sum_c' wt[c'][k] (G[c'][c] - s[c'] s[c] / M)                  (float atomics: the member that accumulates ka (dz^T a) may run beside it)
 * The rank-1 term cancels against ka (dz^T a) wherever dz has a mean: nothing here is rounded to bf16. */
int clite_bn_fold_wgrad_finish(const float* G, const float* asum, int asum_replicas, int asum_stride, const float* coef, const void* wt, int M, int K, int Cin,
                               float* dw, void* stream);


/* ---- ABI v12: the MI projection block of the loss heads (reference loss.py:12-40: LayerNorm(W2 relu(BatchNorm1d(W1 x)) + b2 + Ws x + bs)): everything
 * around its six small GEMMs in four kernels, bf16, M <= 128 rows (clip-lite_amd/csrc/heads_fused.hip). The GEMMs stay clite_gemm_nt launches that
 * ACCUMULATE (ep.atomic, f32) into zeroed workspaces; what the step pays for between the encoders' forward and backward is dependent launches, and these
 * four replace fifteen of them:
 *   clite_mi_block_fwd1  sc = f32 [M][2 U] with columns [0, U) = x W1^T and [U, 2 U) = x Ws^T. Per 32-column slab: z <- bf16(sc[:, :U]); stats <- {sum z,
 *                        sum z^2, 0} of the stored values (one replica); the running statistics updated `updates` times (the reference runs the block twice per
 *                        step on the same statistics); a <- relu(bn(z)) with the batch statistics.
 *   clite_mi_block_fwd2  dxs = f32 [M][U] = a W2^T. Per row: t <- bf16(dxs + b2 + sc[:, U:] + bs); out <- LayerNorm(t) (ln_gamma, ln_beta, ln_eps); ln_stats
 *                        <- {mean, rstd} per row (clite_layernorm_bwd's layout).
 *   clite_mi_block_bwd1  dxs = f32 [M][Fin + U] with columns [Fin, Fin + U) = dtt W2 (and [0, Fin) = dtt Ws). Per 32-column slab: da masked by a > 0,
 *                        BatchNorm1d backward -> dz (bf16); dgamma, dbeta +=; db2, dbs += colsum(dtt).
 *   clite_mi_block_bwd2  after dz W1 has been accumulated onto columns [0, Fin): dx <- bf16(dxs[:, :Fin] + dres).
 * Fin % 16 == 0, U % 16 == 0, U <= 4096. -1 for M > 128 or a missing operand (the caller then takes the generic kernels). */
typedef struct clite_mi_block {
  int32_t M, Fin, U, updates;
  float momentum, eps, ln_eps;
  int32_t reserved;
  const float* bs;          /* f32 [U] or NULL (feature_shortcut.bias) */
  const float* b2;          /* f32 [U] or NULL */
  const float* gamma;       /* BatchNorm1d weight / bias, running statistics (updated in place) */
  const float* beta;
  float* running_mean;
  float* running_var;
  void* z;                  /* bf16 [M][U]  pre-BatchNorm, kept for backward */
  void* a;                  /* bf16 [M][U]  relu(bn(z)) */
  float* stats;             /* f32 [3][U] */
  float* sc;                /* f32 [M][2 U] workspace of the first two products */
  void* t;                  /* bf16 [M][U] */
  void* out;                /* bf16 [M][U] */
  const float* ln_gamma;
  const float* ln_beta;
  float* ln_stats;          /* f32 [M][2] */
  const void* dtt;          /* bf16 [M][U] */
  void* dz;                 /* bf16 [M][U] */
  float* dxs;               /* forward: f32 [M][U] = a W2^T; backward: f32 [M][Fin + U] */
  const void* dres;         /* bf16 [M][Fin] or NULL: added to dx (the prior discriminator's gradient w.r.t. the same features) */
  void* dx;                 /* bf16 [M][Fin] */
  float* dgamma;            /* f32 [U], +=; any of the four may be NULL */
  float* dbeta;
  float* db2;
  float* dbs;
} clite_mi_block;
int clite_mi_block_fwd1(const clite_mi_block* p, void* stream);
int clite_mi_block_fwd2(const clite_mi_block* p, void* stream);
int clite_mi_block_bwd1(const clite_mi_block* p, void* stream);
int clite_mi_block_bwd2(const clite_mi_block* p, void* stream);

/* nn.MaxPool2d(3, stride 2, padding 1) of the ResNet stem; idx holds the window position (0..8) of the first maximum. */
int clite_maxpool3x3s2_fwd(int dtype, const void* x, void* out, uint8_t* idx, int N, int H, int W, int C, void* stream);
int clite_maxpool3x3s2_bwd(int dtype, const void* dout, const uint8_t* idx, void* dx, int N, int H, int W, int C, void* stream);
/* The stem's BatchNorm + ReLU + max-pool in one pass each way (torchvision ResNet conv1 -> bn1 -> relu -> maxpool, reference encoder.py:36-38):
 * pooled / idx = clite_maxpool3x3s2_fwd(clite_bn_apply(y)) without the post-BN tensor ever being stored (p->M = N*H*W, p->relu ignored: ReLU is
 * part of the stem; running statistics updated as in clite_bn_apply); and dy = clite_bn_bwd_apply(clite_bn_bwd_reduce(clite_maxpool3x3s2_bwd(dpool)
 * masked by relu'(bn(y)))) with neither the un-pooled gradient nor the mask stored: dstats (replicated like p->stats, ZERO on entry) receives the
 * two reductions, dgamma / dbeta (optional) +=. Bit-identical to the unfused sequence. */
int clite_stem_bn_pool_fwd(const clite_bn* p, int dtype, const void* y, void* pooled, uint8_t* idx, int N, int H, int W, void* stream);
int clite_stem_bn_pool_bwd(const clite_bn* p, int dtype, const void* dpool, const uint8_t* idx, const void* y, float* dstats, void* dy,
                           float* dgamma, float* dbeta, int N, int H, int W, void* stream);
/* ABI v11. The forward pass that also leaves what the backward reductions of this BatchNorm need in POOLED size: ymax [N][Ho][Wo][C] (optional) = the
 * BatchNorm input y at each window's argmax (an exact copy), and p->relu_bits (optional) = the packed relu' bits of the pooled output. Every other
 * position of a window receives no gradient, so sum dz and sum dz (y - mean) over the un-pooled tensor equal the same sums over (pooled gradient *
 * relu', ymax): the kernel that writes the pooled gradient accumulates them in its epilogue (clite_epilogue.bn_y = ymax, relu_bits,
 * mask_after_residual, colsum; bn_inv_count = 1 / (N H W) of the UN-pooled tensor) and clite_stem_bn_pool_bwd_apply takes them as `dstats` -
 * clite_stem_bn_pool_bwd's second half alone, without the pass over y (205 MB at batch 128) that produced them. (The sums skip one rounding of the
 * unfused sequence - the un-pooled gradient of a pixel that is the argmax of several windows was rounded to the storage type before it was summed.) */
int clite_stem_bn_pool_fwd_ex(const clite_bn* p, int dtype, const void* y, void* pooled, uint8_t* idx, void* ymax, int N, int H, int W, void* stream);
int clite_stem_bn_pool_bwd_apply(const clite_bn* p, int dtype, const void* dpool, const uint8_t* idx, const void* y, const float* dstats, void* dy,
                                 float* dgamma, float* dbeta, int N, int H, int W, void* stream);
/* nn.AdaptiveAvgPool2d((1,1)) + view (reference encoder.py:63-65): [N][HW][C] -> [N][C] */
int clite_avgpool_fwd(int dtype, const void* x, void* out, int N, int HW, int C, void* stream);
int clite_avgpool_bwd(int dtype, const void* dout, void* dx, int N, int HW, int C, void* stream);
/* batch["image"] f32 NCHW [N][3][H][W] -> zero-padded NHWC4 [N][Hp][Wp][4] (the layout the 7x7 stem conv gathers from) */
int clite_image_to_nhwc4(int dtype, const float* img, void* out, int N, int H, int W, int pad, int Hp, int Wp, void* stream);
/* out[n] += sum_m x[m][n]  — bias gradients */
int clite_colsum(int dtype, const void* x, float* out, int M, int N, void* stream);

/* ---- BERT text encoder pieces outside the GEMMs (transformers.BertModel behind reference encoder.py:165-196).
 * Dropout masks are a pure function of (seed, site, element index) so backward regenerates them (csrc/rng.h). */
/* out = dropout(LayerNorm(x)); stats[row] = (mean, rstd) f32. C % 8 == 0, C <= 2048. Also nn.LayerNorm at loss.py:23. */
int clite_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, float eps, void* out, float* stats,
                        int M, int C, float drop_p, uint64_t drop_seed, uint32_t drop_site, void* stream);
/* ABI v11. The same pass (bf16 only) that also leaves the e4m3 copy of `out` for the fp8 linears that read it (BERT's QKV projection and FFN1)
 * and / or records max |out|: fp8_out / fp8_scale / fp8_amax as in clite_epilogue (delayed scaling; any may be NULL, fp8_out needs fp8_scale). */
int clite_layernorm_fwd_q8(int dtype, const void* x, const float* gamma, const float* beta, float eps, void* out, float* stats,
                           int M, int C, float drop_p, uint64_t drop_seed, uint32_t drop_site, uint8_t* fp8_out, const float* fp8_scale, float* fp8_amax,
                           void* stream);
/* dx = LayerNorm backward of dropout_in(dy); dx_masked (optional) = dropout_out(dx); dgamma/dbeta (optional) += .
 * dcolsum (optional, f32 [C]) += column sums of dx_masked (of dx when there is no masked output) as stored: the bias gradient of the
 * nn.Linear whose output this LayerNorm normalised (BertSelfOutput.dense / BertOutput.dense; the token-type row of BertEmbeddings). */
int clite_layernorm_bwd(int dtype, const void* dy, const void* x, const float* stats, const float* gamma, void* dx, void* dx_masked,
                        float* dgamma, float* dbeta, float* dcolsum, int M, int C, float in_p, uint64_t in_seed, uint32_t in_site,
                        float out_p, uint64_t out_seed, uint32_t out_site, void* stream);
/* BertEmbeddings sum: out[row] = word[ids[row]] + pos[row % L] + type[0] (token_type_ids = 0, position_ids = arange(L)) */
int clite_embed_fwd(int dtype, const int64_t* ids, const void* word, const void* pos, const void* type, void* out,
                    int M, int L, int C, int vocab, void* stream);
/* dword[ids[row]] += d[row] (float atomics) for ids[row] != padding_idx; dpos[l] += sum_b d[b*L+l]. Either may be NULL.
 * padding_idx: the row nn.Embedding(padding_idx=...) never accumulates a gradient into (HF BertEmbeddings.word_embeddings: pad_token_id = 0,
 * behind reference encoder.py:163-170); -1 = none. */
int clite_embed_bwd(int dtype, const int64_t* ids, const void* d, float* dword, float* dpos, int M, int L, int C, int vocab, int padding_idx,
                    void* stream);
/* BertSelfAttention core for L <= 32, head size 64: qkv [B*L][3*H*64] (q|k|v), mask int64 [B][L] (1 = attend) or NULL,
 * ctx [B*L][H*64] = dropout(softmax(q k^T / 8 + (1-mask)*finfo.min)) v */
int clite_attention_fwd(int dtype, const void* qkv, const int64_t* mask, void* ctx, int B, int L, int H,
                        float drop_p, uint64_t drop_seed, uint32_t drop_site, void* stream);
int clite_attention_bwd(int dtype, const void* qkv, const int64_t* mask, const void* dctx, void* dqkv, int B, int L, int H,
                        float drop_p, uint64_t drop_seed, uint32_t drop_site, void* stream);

/* BertPooler tanh backward: out = dy * (1 - y^2), n % 8 == 0 */
int clite_tanh_bwd(int dtype, const void* dy, const void* y, void* out, uint64_t n, void* stream);

/* ---- JSD mutual-information loss (reference loss.py). f1 = img_block(image_features), f2 = text_block(text_features),
 * both [B][D], D % 8 == 0, D <= 2048. neg = NULL: negatives are roll-by-one inside the batch, pair (n, n+1 mod B) (loss.py:214-216);
 * otherwise neg[n] (int32, a permutation of 0..B-1) is the row of f2 paired with row n — the cluster hard-negative branch (loss.py:225-252)
 * stacks [batch; hard negatives] and pairs [B/2 + i | (i+1) mod B/2].
 * work: f32 [B][8] per-sample scratch kept for backward. acc: f32 [2] of the step's 8 pre-zeroed accumulators
 * (clite_loss_finalize): acc[0] += mean softplus(-o+) (= -Ej), acc[1] += mean softplus(o-) (= Em). */
int clite_critic_jsd_fwd(int dtype, const void* f1, const void* f2, const float* temperature, int B, int D, const int32_t* neg, float* work,
                         float* acc, void* stream);
/* out[n] = x[n] / max(||x[n]||_2, 1e-12): F.normalize(p=2, dim=-1) of the embedding-extraction / retrieval path that consumes the trained
 * projection heads (reference retrieval.py:108,127; zero_shot.py). [B][D], D % 8 == 0, D <= 2048. */
int clite_l2_normalize(int dtype, const void* x, void* out, int B, int D, void* stream);
int clite_l2_normalize_bwd(int dtype, const void* x, const void* y, const void* dy, void* dx, int B, int D, void* stream);
/* InfoNCE all-pairs variant of the cross-modal term (BASELINE config 4 / SURVEY §8f N4; the reference has no such code): symmetric
 * cross-entropy with diagonal targets over S = exp(temperature) * C, C f32 [B][ld] the cosines of the L2-normalised projections
 * (one clite_gemm_nt). lse_r / lse_c: f32 [B] kept for backward; acc[0] += mean_i(lse_r - S_ii)/2, acc[1] += mean_j(lse_c - S_jj)/2
 * (same slots as the JSD terms, so clite_loss_finalize is shared). */
int clite_infonce_fwd(const float* C, int ld, int B, const float* temperature, float* lse_r, float* lse_c, float* acc, void* stream);
/* dC (call dtype, [B][ldd], ldd % 8 == 0, padding columns zeroed) = dL/dC; dtemp (f32 scalar) += ; scale = (1 - prior_weight). */
int clite_infonce_bwd(int dtype, const float* C, int ld, int B, const float* temperature, const float* lse_r, const float* lse_c,
                      const float* gout, float scale, void* dC, int ldd, float* dtemp, void* stream);
/* gout: device scalar dL/d(total); scale = (1 - prior_weight). df1/df2 in dtype; dtemp (f32 scalar) += . neg as in the forward call,
 * neg_inv its inverse permutation (both NULL = roll by one). */
int clite_critic_jsd_bwd(int dtype, const void* f1, const void* f2, const float* temperature, const float* work, const float* gout, float scale,
                         int B, int D, const int32_t* neg, const int32_t* neg_inv, void* df1, void* df2, float* dtemp, void* stream);
/* PriorDiscriminator.l2 + sigmoid + log terms (loss.py:49-53,189-193) on h1 = relu(l1(relu(l0([u; f])))) stacked [2B][K]:
 * acc += -(mean log D(u) + mean log(1 - D(f))), evaluated as the reference does (log of the f32 sigmoid) when softplus = 0. K % 8 == 0.
 * logit: f32 [2B] kept for backward. With softplus = 1 the same pair of calls is the tail of the `concat` critic GlobalDiscriminator
 * (loss.py:56-68,206-222): rows [0,B) = logits of the positive pairs, acc += mean softplus(-o) (= -Ej); rows [B,2B) = negative pairs,
 * acc += mean softplus(o) (= Em); K = 512. The two forms have the same derivative (clite_prior_tail_bwd). */
int clite_prior_tail_fwd(int dtype, const void* h1, const float* w2, const float* b2, int B, int K, int softplus, float* logit, float* acc,
                         void* stream);
/* dh1 = gradient w.r.t. l1's pre-activation (ReLU mask applied); dw2/db2 (f32) += ; scale = prior_weight. */
int clite_prior_tail_bwd(int dtype, const void* h1, const float* w2, const float* logit, const float* gout, float scale, int B, int K,
                         void* dh1, float* dw2, float* db2, void* stream);
/* acc: f32 [8] = [-Ej, Em | image prior, text prior | visual SSL -Ej, Em | textual SSL -Ej, Em] (loss.py:256-300 for the SSL terms).
 * out: f32 [8]: out[0] = (1-w)*(cross + visual + textual) + w*prior (loss.py:302-305), out[1] = cross = Em - Ej, out[2] = prior,
 * out[3] = visual, out[4] = textual, rest 0. */
int clite_loss_finalize(const float* acc, float prior_weight, float* out, void* stream);
/* out = a + b, n % 8 == 0 elements of `dtype`: feature gradients that arrive from several loss terms (cross-modal + SSL + prior). */
int clite_add(int dtype, const void* a, const void* b, void* out, uint64_t n, void* stream);
/* torch.rand_like replacement for the prior noise (loss.py:189,196): U[0,1) from Philox(seed, site, index). n % 8 == 0. */
int clite_uniform_fill(int dtype, void* out, uint64_t n, uint64_t seed, uint32_t site, void* stream);

/* ---- Update path (reference train.py:221-226, factories.py:464-482, optim/lookahead.py:88-101) over flat f32 buffers. */
typedef struct clite_optim_item {   /* one workgroup's slice of one parameter tensor (never straddles tensors) */
  uint64_t start;                   /* element offset into the flat buffers, multiple of 4 */
  uint32_t count;                   /* elements, multiple of 4 */
  float lr;                         /* base learning rate of the owning tensor's param group */
  float wd;                         /* weight decay of the owning tensor's param group */
  uint32_t reserved;
} clite_optim_item;
/* out (f32 scalar, pre-zeroed) += sum x^2 — the global gradient norm of clip_grad_norm_. Two fixed-order stages through `partials`
 * (device f32[n_partials], n_partials >= 1; 1024 slots use the whole chip): no float atomics, so the value is a pure function of x and every
 * data-parallel rank derives the same clip factor from the same all-reduced gradients (replicas stay bit-identical). */
int clite_sumsq(const float* x, uint64_t n, float* out, float* partials, int n_partials, void* stream);
/* dst[i] = sum over s < slices of src[s * stride + i], i < n, in slice order (stride % 4 == 0, 16-byte aligned pointers). The local step of the
 * mesh gradient exchange that replaces DDP's ring all-reduce (reference train.py:174-178): after an all-to-all every rank holds its chunk of
 * the gradients from all `slices` = world_size ranks and sums them before the all-gather. */
int clite_sum_slices(const float* src, int slices, uint64_t stride, uint64_t n, float* dst, void* stream);
/* hp (device f32[6]): lr multiplier, momentum, max grad norm (<=0 off), lookahead-sync flag, lookahead alpha, grad pre-scale.
 * g' = g*prescale*clip + wd*p; v = mu*v + g'; p -= lr*mult*v; on sync steps p = alpha*p + (1-alpha)*slow, slow = p.
 * g is zeroed; cast_bf16 (optional) receives the bf16 copy of p at the same offsets. */
int clite_sgd_step(float* p, float* g, float* v, float* slow, void* cast_bf16, const clite_optim_item* items, int n_items,
                   const float* hp, const float* sumsq, void* stream);
/* ABI v11. torch.optim.AdamW (reference factories.py:439: OPTIMIZER_NAME "adamw", torch's defaults) in the same one-pass form, with the same clipping,
 * Lookahead synchronisation, gradient zeroing and bf16 copy. hp (device f32[12]): as clite_sgd_step, with [1] = beta1, [6] = beta2, [7] = eps,
 * [8] = 1 - beta1^t, [9] = 1 - beta2^t of this step, [10] = 1 - beta1, [11] = 1 - beta2 (formed in double by the host). p *= 1 - lr wd; m += (1 - beta1)(g' - m); v2 = beta2 v2 + (1 - beta2) g'^2;
 * p -= lr / hp[8] * m / (sqrt(v2) / sqrt(hp[9]) + eps), g' = g * prescale * clip. */
int clite_adamw_step(float* p, float* g, float* m, float* v2, float* slow, void* cast_bf16, const clite_optim_item* items, int n_items,
                     const float* hp, const float* sumsq, void* stream);
int clite_cast_bf16(const float* src, void* dst, uint64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CLITE_H */
