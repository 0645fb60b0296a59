// Internal interface between the launchers of gemm.hip and the wide-K 8-wave kernels of gemm_wide.hip (separate translation units so
// the two instantiation sets compile in parallel).
#ifndef CLITE_WIDE_API_H
#define CLITE_WIDE_API_H
#include "igemm.h"

namespace clite {

enum { WOP_KC = 0, WOP_KC_DGRAD = 1, WOP_XC_STRIDED = 2, WOP_XC_GATHER = 3 };
struct WideOperand {      // one GEMM operand as the loaders of igemm_wide.h / igemm_dma.h describe it
  int kind;
  const void* ptr;
  uint32_t bytes;
  ConvGeom g;             // WOP_KC, WOP_KC_DGRAD, WOP_XC_GATHER
  int ld, Cx, Ck, RS;     // WOP_XC_STRIDED
};
constexpr int WIDE_NOT_TAKEN = -1000;
// bf16 only. Returns WIDE_NOT_TAKEN when the launch stays on the 4-wave kernels (shape, policy), else the HIP status of the launch.
int launch_wide(const WideOperand& a, const WideOperand& b, const clite_epilogue& ep, const RowMap& rm, int M, int N, int Ktot, int splits, hipStream_t st);
// conv_patch.hip: the patch-resident 3 x 3 / stride 1 kernel for 64 -> 64 channels (bf16), forward (w = [K][3][3][C]) or input gradient on the
// transposed weights (dgrad: x = dy, w = [C][3][3][K]). Returns WIDE_NOT_TAKEN for every launch it does not cover.
int launch_conv3x3_patch(const void* x, const void* w, const clite_conv& c, const clite_epilogue& ep, bool dgrad, hipStream_t st);
// ... and its weight gradient (dw f32 [K][3][3][C] += ...) through a workspace of conv3x3_wgrad_patch_workspace() bytes (per-workgroup partial sums)
size_t conv3x3_wgrad_patch_workspace();
int launch_conv3x3_wgrad_patch(const void* dy, const void* x, const clite_conv& c, float* dw, void* ws, size_t ws_bytes, hipStream_t st);
// the stem's weight gradient on the same scheme (dw f32 [64][7][7][3] +=; the same workspace serves)
// ... and fused with bn1's backward (the un-pooled gradient formed in LDS from the pooled gradient, the window indices and conv1's output)

int launch_stem_wgrad_patch(const void* dy, const void* xpad, int N, int Hp, int Wp, int Ho, int Wo, float* dw, void* ws, size_t ws_bytes, hipStream_t st);
// fold_dgrad.hip: the folded BatchNorm backward's input gradient for K = 256, Cin = 64 (the 56 x 56 stage) as a streaming kernel — whole-row K tiles, weights
// resident in LDS. Returns WIDE_NOT_TAKEN for every launch it does not cover (shape, epilogue form, deterministic mode, a forced tile policy).
int launch_fold_dgrad_rows(const void* pair, const void* w2, int M, int K, int Cin, const clite_epilogue& ep, hipStream_t st);
// the tile policy set by clite_set_tile_policy (gemm_wide.hip): 0 = automatic; the forced forms keep every launch on the kernel family they name
int tile_policy_value();

}  // namespace clite
#endif
