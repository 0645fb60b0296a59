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
 *   - entry points are re-entrant and hold no global mutable state (forward runs on the Python main
 *     thread, backward on the autograd engine thread)
 */
#ifndef CLITE_H
#define CLITE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLITE_ABI_VERSION 1
int clite_abi_version(void);

/* storage type of activations / weights handed to a kernel. CLITE_F32 selects the exact-f32 parity mode
 * (v_mfma_f32_32x32x2_f32, 1/16 of the bf16 MFMA rate). Accumulation is f32 in both. */
enum { CLITE_BF16 = 0, CLITE_F32 = 1 };

enum { CLITE_ACT_NONE = 0, CLITE_ACT_RELU = 1, CLITE_ACT_GELU = 2, CLITE_ACT_TANH = 3 };

/* Fused GEMM epilogue. v = alpha*acc + bias; preact <- v; v = act(v); v *= act'(dact_aux);
 * v = dropout(v); v += residual; out <- v; colsum += (sum, sum of squares) of the stored values. */
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
} clite_epilogue;

/* NHWC convolution problem. x: [N][H][W][C], w: [K][R][S][C], y: [N][Ho][Wo][K], all of `dtype`. */
typedef struct clite_conv {
  int32_t dtype;
  int32_t N, H, W, C;
  int32_t K, R, S;
  int32_t stride, pad;
  int32_t Ho, Wo;
} clite_conv;

/* C[M,N] = A[M,K] * B[N,K]^T  — nn.Linear forward (reference loss.py:16-22,46-48; HF BertModel linears
 * behind encoder.py:193). K % 8 == 0, N % 8 == 0. */
int clite_gemm_nt(const void* A, const void* B, int M, int N, int K, int dtype, const clite_epilogue* ep, void* stream);
/* C[M,N] = A[M,K] * B[K,N]       — nn.Linear input gradient (autograd of the same call sites). */
int clite_gemm_nn(const void* A, const void* B, int M, int N, int K, int dtype, const clite_epilogue* ep, void* stream);
/* C[M,N] (+)= A[K,M]^T * B[K,N]  — nn.Linear weight gradient; ep->atomic selects f32 accumulation. */
int clite_gemm_tn(const void* A, const void* B, int M, int N, int K, int dtype, const clite_epilogue* ep, void* stream);

/* y = conv(x, w): torchvision ResNet nn.Conv2d forward (reference encoder.py:36-38,63; block arithmetic as
 * restated in model_zoo/resnet.py:60-100). Epilogue applies to y viewed as [N*Ho*Wo][K]. */
int clite_conv_fwd(const void* x, const void* w, const clite_conv* cv, const clite_epilogue* ep, void* stream);
/* dx = conv_transpose(dy, w): autograd of the same call. Epilogue applies to dx viewed as [N*H*W][C]. */
int clite_conv_dgrad(const void* dy, const void* w, const clite_conv* cv, const clite_epilogue* ep, void* stream);
/* dw[K][R][S][C] (f32) += dy^T * im2col(x): autograd of the same call; float-atomic split-K. */
int clite_conv_wgrad(const void* dy, const void* x, const clite_conv* cv, float* dw, void* stream);

/* ResNet stem conv1 = nn.Conv2d(3, 64, 7, stride 2, padding 3, bias=False) (torchvision, reference encoder.py:36-38) on the
 * pre-padded NHWC4 image from clite_image_to_nhwc4 (Hp >= 2*(Ho-1)+7, Wp >= 2*(Wo-1)+8, Wp even). wv is the weight packed
 * as [64][7][8][4] (clite_stem_pack from f32 [64][7][7][3]); clite_stem_unpack_grad folds the packed gradient back (+=). */
int clite_stem_fwd(const void* xpad, const void* wv, int dtype, int N, int Hp, int Wp, int Ho, int Wo, const clite_epilogue* ep, void* stream);
int clite_stem_wgrad(const void* dy, const void* xpad, int dtype, int N, int Hp, int Wp, int Ho, int Wo, float* dwv, void* stream);
int clite_stem_pack(const float* w, void* wv, int dtype, void* stream);
int clite_stem_unpack_grad(const float* dwv, float* dw, void* stream);

/* ---- BatchNorm2d / BatchNorm1d, train-mode batch statistics (torchvision ResNet BN behind reference
 * encoder.py:36-65; nn.BatchNorm1d at loss.py:18). Tensors are [M][C] (NHWC flattened), C % 8 == 0, C/8 divides 256.
 * `stats` = per-channel (sum, sum of squares) [2][C] f32 over the M rows, produced by the conv/GEMM epilogue
 * (clite_epilogue.colsum). Biased variance normalises; unbiased variance goes into running_var. */
typedef struct clite_bn {
  int32_t M, C;
  const float* stats;        /* [2][C] */
  const float* gamma;        /* [C] */
  const float* beta;         /* [C] */
  float* running_mean;       /* [C] */
  float* running_var;        /* [C] */
  int32_t training;          /* 1: batch statistics; 0: running statistics (eval) */
  int32_t update_running;    /* 1: workgroup 0 updates running_* with `momentum` */
  float momentum, eps;
  int32_t relu;              /* apply ReLU at the end */
  /* optional second BN applied to the residual operand (the 1x1/stride downsample branch of a block) */
  const float* res_stats;
  const float* res_gamma;
  const float* res_beta;
  float* res_running_mean;
  float* res_running_var;
} clite_bn;

/* out = relu?( bn(y) + [res | bn_res(res)] ) */
int clite_bn_apply(const clite_bn* p, int dtype, const void* y, const void* res, void* out, void* stream);
/* dstats[0][c] += sum dz, dstats[1][c] += sum dz*y, dz = dout * (mask > 0) (mask NULL: dz = dout). dstats pre-zeroed. */
int clite_bn_bwd_reduce(int dtype, const void* dout, const void* mask, const void* y, float* dstats, int M, int C, void* stream);
/* dy = BN backward of dz through batch statistics; dz (optional) <- masked dout; dgamma/dbeta (optional) += . */
int clite_bn_bwd_apply(const clite_bn* p, int dtype, const void* dout, const void* mask, const void* y, const float* dstats,
                       void* dy, void* dz, float* dgamma, float* dbeta, void* stream);

/* nn.MaxPool2d(3, stride 2, padding 1) of the ResNet stem; idx holds the window position (0..8) of the first maximum. */
int clite_maxpool3x3s2_fwd(int dtype, const void* x, void* out, uint8_t* idx, int N, int H, int W, int C, void* stream);
int clite_maxpool3x3s2_bwd(int dtype, const void* dout, const uint8_t* idx, void* dx, int N, int H, int W, int C, void* stream);
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
/* dx = LayerNorm backward of dropout_in(dy); dx_masked (optional) = dropout_out(dx); dgamma/dbeta (optional) += . */
int clite_layernorm_bwd(int dtype, const void* dy, const void* x, const float* stats, const float* gamma, void* dx, void* dx_masked,
                        float* dgamma, float* dbeta, int M, int C, float in_p, uint64_t in_seed, uint32_t in_site,
                        float out_p, uint64_t out_seed, uint32_t out_site, void* stream);
/* BertEmbeddings sum: out[row] = word[ids[row]] + pos[row % L] + type[0] (token_type_ids = 0, position_ids = arange(L)) */
int clite_embed_fwd(int dtype, const int64_t* ids, const void* word, const void* pos, const void* type, void* out,
                    int M, int L, int C, int vocab, void* stream);
/* dword[ids[row]] += d[row] (float atomics); dpos[l] += sum_b d[b*L+l]. Either may be NULL. */
int clite_embed_bwd(int dtype, const int64_t* ids, const void* d, float* dword, float* dpos, int M, int L, int C, int vocab, void* stream);
/* BertSelfAttention core for L <= 32, head size 64: qkv [B*L][3*H*64] (q|k|v), mask int64 [B][L] (1 = attend) or NULL,
 * ctx [B*L][H*64] = dropout(softmax(q k^T / 8 + (1-mask)*finfo.min)) v */
int clite_attention_fwd(int dtype, const void* qkv, const int64_t* mask, void* ctx, int B, int L, int H,
                        float drop_p, uint64_t drop_seed, uint32_t drop_site, void* stream);
int clite_attention_bwd(int dtype, const void* qkv, const int64_t* mask, const void* dctx, void* dqkv, int B, int L, int H,
                        float drop_p, uint64_t drop_seed, uint32_t drop_site, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CLITE_H */
