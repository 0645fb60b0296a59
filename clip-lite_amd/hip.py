"""ctypes binding of libclite_hip.so (C ABI: include/clite.h) — the only route from the Python host layer to device math.

There is deliberately no fallback: if the gfx950 library is missing or a kernel returns an error, the caller gets a
RuntimeError. PyTorch is used for device memory, streams and torch.distributed only.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CLITE_HIP_LIB") or os.path.join(_HERE, "lib", "libclite_hip.so")     # override: diagnostic builds only
ABI_VERSION = 12

BF16, F32 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU, ACT_TANH = 0, 1, 2, 3
DACT_RELU, DACT_GELU, DACT_TANH = 1, 2, 3
SEED_INDIRECT = 0x80000000   # OR-ed into a dropout site: the seed argument is the device address of a uint64 (CLITE_SEED_INDIRECT)

TORCH_DTYPE = {BF16: torch.bfloat16, F32: torch.float32}


class Epilogue(C.Structure):
    _fields_ = [
        ("out", C.c_void_p), ("ldc", C.c_int32), ("out_f32", C.c_int32), ("atomic", C.c_int32),
        ("alpha", C.c_float), ("bias", C.c_void_p), ("act", C.c_int32), ("preact", C.c_void_p),
        ("dact_aux", C.c_void_p), ("dact", C.c_int32), ("drop_p", C.c_float), ("drop_seed", C.c_uint64),
        ("drop_site", C.c_uint32), ("residual", C.c_void_p), ("colsum", C.c_void_p),
        ("colsum_replicas", C.c_int32), ("colsum_stride", C.c_int32), ("colsum_rows", C.c_int32),
        ("bn_y", C.c_void_p), ("bn_stats", C.c_void_p), ("bn_replicas", C.c_int32), ("bn_rstride", C.c_int32),
        ("bn_inv_count", C.c_float), ("mask_after_residual", C.c_int32), ("relu_bits", C.c_void_p), ("splitk_ws", C.c_void_p), ("residual_subsample", C.c_int32),
        ("fp8_out", C.c_void_p), ("fp8_scale", C.c_void_p), ("fp8_amax", C.c_void_p),
    ]


class Conv(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dtype", "N", "H", "W", "C", "K", "R", "S", "stride", "pad", "Ho", "Wo")]


class Bn(C.Structure):
    _fields_ = [
        ("M", C.c_int32), ("C", C.c_int32), ("stats", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("training", C.c_int32),
        ("update_running", C.c_int32), ("momentum", C.c_float), ("eps", C.c_float), ("relu", C.c_int32), ("replicas", C.c_int32), ("rstride", C.c_int32), ("centered", C.c_int32),
        ("res_stats", C.c_void_p), ("res_gamma", C.c_void_p), ("res_beta", C.c_void_p),
        ("res_running_mean", C.c_void_p), ("res_running_var", C.c_void_p), ("relu_bits", C.c_void_p),
        ("fp8_out", C.c_void_p), ("fp8_scale", C.c_void_p), ("fp8_amax", C.c_void_p),
        ("out_sum", C.c_void_p), ("out_sum_replicas", C.c_int32), ("out_sum_stride", C.c_int32),
    ]


class MiBlock(C.Structure):
    _fields_ = [("M", C.c_int32), ("Fin", C.c_int32), ("U", C.c_int32), ("updates", C.c_int32), ("momentum", C.c_float), ("eps", C.c_float), ("ln_eps", C.c_float),
                ("reserved", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("bs", "b2", "gamma", "beta", "running_mean", "running_var", "z", "a", "stats", "sc", "t", "out", "ln_gamma", "ln_beta",
                                          "ln_stats", "dtt", "dz", "dxs", "dres", "dx", "dgamma", "dbeta", "db2", "dbs")]


class WgradItem(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_void_p), ("b", C.c_void_p), ("out", C.c_void_p), ("cv", Conv),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32),
                ("a_scales", C.c_void_p), ("b_scales", C.c_void_p), ("row_scale", C.c_void_p)]


class Fp8Item(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("numel", C.c_uint64)]


class TransposeItem(C.Structure):
    _fields_ = [("src_off", C.c_uint64), ("dst_off", C.c_uint64)] + [(n, C.c_uint32) for n in
                ("rows", "cols", "src_ld", "dst_ld", "batch", "src_bstride", "dst_bstride", "first_tile")]


class OptimItem(C.Structure):
    _fields_ = [("start", C.c_uint64), ("count", C.c_uint32), ("lr", C.c_float), ("wd", C.c_float), ("reserved", C.c_uint32)]


_V, _I, _F, _U64, _U32 = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_uint32
_SIGNATURES = {
    "clite_abi_version": [],
    "clite_set_deterministic": [_I],
    "clite_get_deterministic": [],
    "clite_set_tile_policy": [_I],
    "clite_set_f32_split": [_I],
    "clite_get_f32_split": [],
    "clite_gemm_nt": [_V, _I, _V, _I, _I, _I, _I, _I, _V, _V],
    "clite_gemm_nn": [_V, _I, _V, _I, _I, _I, _I, _I, _V, _V],
    "clite_gemm_tn": [_V, _I, _V, _I, _I, _I, _I, _I, _V, _V],
    "clite_conv_fwd": [_V, _V, _V, _V, _V],
    "clite_conv_dgrad": [_V, _V, _V, _V, _V],
    "clite_conv_dgrad_s2class": [_V, _V, _V, _I, _I, _V, _V],
    "clite_conv_dgrad_wt": [_V, _V, _V, _V, _V],
    "clite_conv_dgrad_s2class_wt": [_V, _V, _V, _I, _I, _V, _V],
    "clite_mi_block_fwd1": [_V, _V],
    "clite_mi_block_fwd2": [_V, _V],
    "clite_mi_block_bwd1": [_V, _V],
    "clite_mi_block_bwd2": [_V, _V],
    "clite_bn_fold_prepare": [_V, _V, _V, _I, _V, _V, _V, _V, _V, _V],
    "clite_conv_dgrad_bnfold": [_V, _V, _I, _I, _I, _V, _V],
    "clite_bn_fold_wgrad_finish": [_V, _V, _I, _I, _V, _V, _I, _I, _I, _V, _V],
    "clite_transpose_weights": [_V, _V, _V, _I, _U32, _V],
    "clite_conv_wgrad": [_V, _V, _V, _V, _V],
    "clite_conv_wgrad_patch_workspace": [_V],
    "clite_conv_wgrad_patch": [_V, _V, _V, _V, _V, _U64, _V],
    "clite_wgrad_group": [_I, _V, _I, _V, _V, _U64, _V],
    "clite_fp8_quantize": [_I, _V, _U64, _V, _V, _V, _V],
    "clite_gemm_nt_fp8": [_V, _I, _V, _I, _I, _I, _I, _V, _V, _V, _V],
    "clite_conv_fwd_fp8": [_V, _V, _V, _V, _V, _V, _V],
    "clite_conv_dgrad_fp8": [_V, _V, _V, _V, _V, _V, _V],
    "clite_fp8_scale_update": [_V, _V, _I, _V],
    "clite_fp8_quantize_group": [_V, _V, _V, _I, _I, _V, _V, _V, _V, _V],
    "clite_wgrad_group_workspace": [_I, C.c_int64, _V],
    "clite_stem_fwd": [_V, _V, _I, _I, _I, _I, _I, _I, _V, _V],
    "clite_stem_wgrad": [_V, _V, _I, _I, _I, _I, _I, _I, _V, _V],
    "clite_stem_wgrad_patch": [_V, _V, _I, _I, _I, _I, _I, _I, _V, _V, _U64, _V],
    "clite_stem_pack": [_V, _V, _I, _V],
    "clite_stem_unpack_grad": [_V, _V, _V],
    "clite_bn_apply": [_V, _I, _V, _V, _V, _V],
    "clite_bn_centered_var": [_I, _V, _V, _I, _I, _I, _I, _V],
    "clite_bn_bwd_reduce": [_I, _V, _V, _V, _V, _V, _V, _I, _I, _I, _I, _V],
    "clite_bn_bwd_apply": [_V, _I, _V, _V, _V, _V, _V, _V, _V, _V, _V, _V],
    "clite_maxpool3x3s2_fwd": [_I, _V, _V, _V, _I, _I, _I, _I, _V],
    "clite_maxpool3x3s2_bwd": [_I, _V, _V, _V, _I, _I, _I, _I, _V],
    "clite_stem_bn_pool_fwd": [_V, _I, _V, _V, _V, _I, _I, _I, _V],
    "clite_stem_bn_pool_bwd": [_V, _I, _V, _V, _V, _V, _V, _V, _V, _I, _I, _I, _V],
    "clite_stem_bn_pool_fwd_ex": [_V, _I, _V, _V, _V, _V, _I, _I, _I, _V],
    "clite_stem_bn_pool_bwd_apply": [_V, _I, _V, _V, _V, _V, _V, _V, _V, _I, _I, _I, _V],
    "clite_avgpool_fwd": [_I, _V, _V, _I, _I, _I, _V],
    "clite_avgpool_bwd": [_I, _V, _V, _I, _I, _I, _V],
    "clite_image_to_nhwc4": [_I, _V, _V, _I, _I, _I, _I, _I, _I, _V],
    "clite_colsum": [_I, _V, _V, _I, _I, _V],
    "clite_layernorm_fwd": [_I, _V, _V, _V, _F, _V, _V, _I, _I, _F, _U64, _U32, _V],
    "clite_layernorm_fwd_q8": [_I, _V, _V, _V, _F, _V, _V, _I, _I, _F, _U64, _U32, _V, _V, _V, _V],
    "clite_layernorm_bwd": [_I, _V, _V, _V, _V, _V, _V, _V, _V, _V, _I, _I, _F, _U64, _U32, _F, _U64, _U32, _V],
    "clite_embed_fwd": [_I, _V, _V, _V, _V, _V, _I, _I, _I, _I, _V],
    "clite_embed_bwd": [_I, _V, _V, _V, _V, _I, _I, _I, _I, _I, _V],
    "clite_attention_fwd": [_I, _V, _V, _V, _I, _I, _I, _F, _U64, _U32, _V],
    "clite_attention_bwd": [_I, _V, _V, _V, _V, _I, _I, _I, _F, _U64, _U32, _V],
    "clite_tanh_bwd": [_I, _V, _V, _V, _U64, _V],
    "clite_critic_jsd_fwd": [_I, _V, _V, _V, _I, _I, _V, _V, _V, _V],
    "clite_l2_normalize": [_I, _V, _V, _I, _I, _V],
    "clite_l2_normalize_bwd": [_I, _V, _V, _V, _V, _I, _I, _V],
    "clite_infonce_fwd": [_V, _I, _I, _V, _V, _V, _V, _V],
    "clite_infonce_bwd": [_I, _V, _I, _I, _V, _V, _V, _V, _F, _V, _I, _V, _V],
    "clite_critic_jsd_bwd": [_I, _V, _V, _V, _V, _V, _F, _I, _I, _V, _V, _V, _V, _V, _V],
    "clite_prior_tail_fwd": [_I, _V, _V, _V, _I, _I, _I, _V, _V, _V],
    "clite_prior_tail_bwd": [_I, _V, _V, _V, _V, _F, _I, _I, _V, _V, _V, _V],
    "clite_loss_finalize": [_V, _F, _V, _V],
    "clite_add": [_I, _V, _V, _V, _U64, _V],
    "clite_uniform_fill": [_I, _V, _U64, _U64, _U32, _V],
    "clite_sumsq": [_V, _U64, _V, _V, _I, _V],
    "clite_sum_slices": [_V, _I, _U64, _U64, _V, _V],
    "clite_sgd_step": [_V, _V, _V, _V, _V, _V, _I, _V, _V, _V],
    "clite_adamw_step": [_V, _V, _V, _V, _V, _V, _V, _I, _V, _V, _V],
    "clite_cast_bf16": [_V, _V, _U64, _V],
}

_lib = None
_allow_host_tensors = False   # set only by tests that inject the wave-simulator build of the same sources


def _bind(l):
    for name, args in _SIGNATURES.items():
        fn = getattr(l, name)        # AttributeError here = the library does not export what include/clite.h declares
        fn.argtypes = args
        fn.restype = C.c_int
    return l


def lib():
    """Load the kernel library once; fail loudly when it is absent (no CPU or eager fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"clip_lite_amd: {LIB_PATH} not found. Build it with `make hip` (or __graft_entry__.build()); "
                "this package has no fallback path.")
        l = _bind(C.CDLL(LIB_PATH))
        if l.clite_abi_version() != ABI_VERSION:
            raise RuntimeError("clip_lite_amd: libclite_hip.so ABI version mismatch; rebuild with `make hip`")
        _lib = l
        if os.environ.get("CLITE_DETERMINISTIC", "0") not in ("", "0"):
            l.clite_set_deterministic(1)
        if os.environ.get("CLITE_F32_SPLIT", "0") not in ("", "0"):
            l.clite_set_f32_split(1)
    return _lib


def set_deterministic(on=True):
    """Deterministic-reduction mode of the kernel library (include/clite.h: clite_set_deterministic): bit-identical runs, several times
    slower. Also switched on by the environment variable CLITE_DETERMINISTIC=1."""
    lib().clite_set_deterministic(int(bool(on)))


def is_deterministic():
    return bool(lib().clite_get_deterministic())


def set_f32_split(on=True):
    """Split-bf16 form of the exact-f32 mode's matrix products (include/clite.h: clite_set_f32_split): f32 storage, three bf16 MFMAs per product,
    ~2^-17 relative error per product. Also switched on by the environment variable CLITE_F32_SPLIT=1."""
    lib().clite_set_f32_split(int(bool(on)))


def is_f32_split():
    return bool(lib().clite_get_f32_split())


TILE_AUTO, TILE_W128, TILE_W256x128, TILE_W256, TILE_NARROW = 0, 1, 2, 3, 4


def set_tile_policy(policy=TILE_AUTO):
    """Tile-shape policy of the bf16 GEMM / conv launchers (include/clite.h: clite_set_tile_policy); the forced forms are for parity tests."""
    check(lib().clite_set_tile_policy(int(policy)), "set_tile_policy")


def exported_symbols():
    return sorted(_SIGNATURES)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(t=None):
    """hipStream_t of the current stream of t's device (one call per kernel launch: the raw accessor is ~10x cheaper than building a
    torch.cuda.Stream object each time)."""
    if t is not None and not t.is_cuda:
        return None
    if _raw_stream is not None:
        return _raw_stream(t.device.index if t is not None and t.device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def p(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda and not _allow_host_tensors:
        raise RuntimeError("clip_lite_amd: kernels need CUDA/HIP device tensors; this package has no CPU path")
    return t.data_ptr()


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"clip_lite_amd: {what} failed with code {rc}")


class Stats:
    """Replicated per-channel accumulator [R][3][C] f32 (rows: sum, sum of squares, centered sum of squares), zero-initialised."""
    __slots__ = ("t", "R", "C")

    def __init__(self, t, R, Cc):
        self.t, self.R, self.C = t, R, Cc

    @property
    def rstride(self):
        return 3 * self.C


def epilogue(out, ldc=None, atomic=False, alpha=1.0, bias=None, act=ACT_NONE, preact=None, dact_aux=None, dact=0,
             drop=None, residual=None, colsum=None, out_f32=None, bn=None, mask_after_residual=False, ws=None, relu_bits=None, colsum_rows=0,
             residual_subsample=0, fp8=None):
    """fp8 = (q uint8 [M][ldc] | None, scales f32[2] | None, amax slot | None): clite_gemm_nt_fp8 only - the e4m3 copy of `out` for the next fp8
    GEMM at a delayed scale, and this call's max |out| (clite_epilogue.fp8_*).
    bn = (y, stats: Stats, rows): accumulate the BatchNorm-backward reductions (sum v, sum v*(y - mean)) into `colsum`.
    relu_bits: the relu' mask of that form as packed bits (uint8 [M][ldc / 8], written by bn_apply) instead of dact_aux."""
    ep = Epilogue()
    ep.out = p(out)
    ep.ldc = ldc if ldc is not None else out.shape[-1]
    ep.out_f32 = int(out.dtype == torch.float32) if out_f32 is None else int(out_f32)
    ep.atomic = int(atomic)
    ep.alpha = alpha
    ep.bias = p(bias)
    ep.act = act
    ep.preact = p(preact)
    ep.dact_aux = p(dact_aux)
    ep.dact = dact
    if drop is not None and drop[0] > 0.0:
        ep.drop_p, ep.drop_seed, ep.drop_site = drop
    ep.residual = p(residual)
    if isinstance(colsum, Stats):
        ep.colsum, ep.colsum_replicas, ep.colsum_stride = p(colsum.t), colsum.R, colsum.rstride
    else:
        ep.colsum = p(colsum)
    ep.colsum_rows = colsum_rows      # 1: column sums only (a bias gradient accumulated by the GEMM that produces the gradient tensor)
    if bn is not None:
        y, st, rows = bn
        ep.bn_y, ep.bn_stats, ep.bn_replicas, ep.bn_rstride, ep.bn_inv_count = p(y), p(st.t), st.R, st.rstride, 1.0 / rows
    ep.mask_after_residual = int(mask_after_residual)
    ep.relu_bits = p(relu_bits)
    ep.residual_subsample = residual_subsample      # 2: `residual` is the compact [N][H/2][W/2][ldc] gradient of a stride-2 shortcut (clite_epilogue.residual_subsample)
    if fp8 is not None:
        ep.fp8_out, ep.fp8_scale, ep.fp8_amax = p(fp8[0]), p(fp8[1]), p(fp8[2])
    ep.splitk_ws = p(ws)        # zeroed f32 [M][N] workspace: allows split-K for GEMMs of few output tiles (clite_epilogue.splitk_ws)
    return ep


# ------------------------------------------------------------------------------------------------ GEMM / conv
def gemm_nt(dt, A, B, M, N, K, ep, lda=None, ldb=None):
    check(lib().clite_gemm_nt(p(A), lda or K, p(B), ldb or K, M, N, K, dt, C.byref(ep), stream_ptr(A)), "gemm_nt")


def gemm_nn(dt, A, B, M, N, K, ep, lda=None, ldb=None):
    check(lib().clite_gemm_nn(p(A), lda or K, p(B), ldb or N, M, N, K, dt, C.byref(ep), stream_ptr(A)), "gemm_nn")


def gemm_tn(dt, A, B, M, N, K, ep, lda=None, ldb=None):
    check(lib().clite_gemm_tn(p(A), lda or M, p(B), ldb or N, M, N, K, dt, C.byref(ep), stream_ptr(A)), "gemm_tn")


def conv_desc(dt, N, H, W, Cin, K, R, S, stride, pad):
    Ho = (H + 2 * pad - R) // stride + 1
    Wo = (W + 2 * pad - S) // stride + 1
    return Conv(dt, N, H, W, Cin, K, R, S, stride, pad, Ho, Wo)


def conv_fwd(x, w, cv, ep):
    check(lib().clite_conv_fwd(p(x), p(w), C.byref(cv), C.byref(ep), stream_ptr(x)), "conv_fwd")


def conv_dgrad(dy, w, cv, ep, wt=False):
    """wt: `w` is the transposed copy [C][R][S][K] (Arena.wt): both GEMM operands k-contiguous (clite_conv_dgrad_wt)."""
    fn = lib().clite_conv_dgrad_wt if wt else lib().clite_conv_dgrad
    check(fn(p(dy), p(w), C.byref(cv), C.byref(ep), stream_ptr(dy)), "conv_dgrad")


def mi_block(**kw):
    """clite_mi_block with the given fields (tensors -> device pointers)."""
    b = MiBlock()
    for k, v in kw.items():
        setattr(b, k, p(v) if torch.is_tensor(v) else v)
    return b


def mi_block_call(which, b, like):
    """which: "fwd1" | "fwd2" | "bwd1" | "bwd2" (include/clite.h: clite_mi_block_*)."""
    check(getattr(lib(), "clite_mi_block_" + which)(C.byref(b), stream_ptr(like)), "mi_block_" + which)


class BnFold:
    """What clite_bn_fold_prepare leaves for the folded BatchNorm backward of one 1 x 1 conv -> BatchNorm unit (include/clite.h, ABI v12)."""
    __slots__ = ("w2", "bias", "coef", "K", "Cin")


def bn_fold_prepare(desc, dstats, wt, Cin, dgamma, dbeta):
    """desc: the BatchNorm (hip.bn_desc, training statistics); dstats: its two backward reductions (Stats, same replica layout); wt: the bf16 TRANSPOSED
    weights [Cin][K] of the 1 x 1 convolution in front of it. Accumulates dgamma / dbeta."""
    K = desc.C
    f = BnFold()
    f.K, f.Cin = K, Cin
    dev = wt.device
    f.w2 = torch.empty(Cin, 2, K, dtype=torch.bfloat16, device=dev)
    f.bias = torch.empty(Cin, dtype=torch.float32, device=dev)
    f.coef = torch.empty(3, K, dtype=torch.float32, device=dev)
    assert dstats.R == desc.replicas and dstats.rstride == desc.rstride
    check(lib().clite_bn_fold_prepare(C.byref(desc), p(dstats.t), p(wt), Cin, p(f.w2), p(f.bias), p(f.coef), p(dgamma), p(dbeta), stream_ptr(wt)), "bn_fold_prepare")
    return f


def conv_dgrad_bnfold(pair, w2, M, K, Cin, ep):
    """pair: bf16 [2][M][K] (slot 0 = dz, slot 1 = y); ep: the BatchNorm-backward form (relu_bits, bn, colsum) with bias = BnFold.bias."""
    assert pair.is_contiguous() and tuple(pair.shape) == (2, M, K)
    check(lib().clite_conv_dgrad_bnfold(p(pair), p(w2), M, K, Cin, C.byref(ep), stream_ptr(pair)), "conv_dgrad_bnfold")


def bn_fold_wgrad_finish(G, asum, coef, wt, M, K, Cin, dw):
    """asum: Stats whose row 0 holds the column sums of the activation (bn_desc(out_sum=...) of the bn_apply that wrote it)."""
    check(lib().clite_bn_fold_wgrad_finish(p(G), p(asum.t), asum.R, asum.rstride, p(coef), p(wt), M, K, Cin, p(dw), stream_ptr(G)), "bn_fold_wgrad_finish")


def transpose_weights(src, dst, items_dev, n_items, total_tiles):
    check(lib().clite_transpose_weights(p(src), p(dst), p(items_dev), n_items, total_tiles, stream_ptr(src)), "transpose_weights")


def s2_class_weights(w):
    """The four tap subsets of a [K][3][3][C] weight that the input-parity classes of a stride-2 dgrad use, class (ph, pw) at index 2 * ph + pw.
    They depend on the weights only, so a captured step makes them early on its side stream instead of in front of each class's GEMM."""
    return [w[:, (ph + 1) & 1::2, (pw + 1) & 1::2, :].contiguous() for ph in (0, 1) for pw in (0, 1)]


def conv_dgrad_s2(dy, w, cv, make_ep, wsubs=None, wt=False):
    """dgrad of a 3x3 / stride-2 / pad-1 conv as its four input-parity classes (a quarter of the MACs of the gathered form).
    `make_ep()` builds the epilogue (called once per class: the classes write disjoint rows of the same output). wt: `w` / `wsubs` are
    (slices of) the transposed weight [C][R][S][K] — the same tap slicing applies to axes 1 and 2 of either layout."""
    if wsubs is None:
        wsubs = s2_class_weights(w)
    fn = lib().clite_conv_dgrad_s2class_wt if wt else lib().clite_conv_dgrad_s2class
    for ph in (0, 1):
        for pw in (0, 1):
            ep = make_ep()
            check(fn(p(dy), p(wsubs[2 * ph + pw]), C.byref(cv), ph, pw, C.byref(ep), stream_ptr(dy)), "conv_dgrad_s2class")


def s2_classes_ok(cv):
    return cv.R == 3 and cv.S == 3 and cv.stride == 2 and cv.pad == 1 and cv.H % 2 == 0 and cv.W % 2 == 0 and cv.C % 64 == 0 and cv.K % 64 == 0


def conv_wgrad(dy, x, cv, dw):
    check(lib().clite_conv_wgrad(p(dy), p(x), C.byref(cv), p(dw), stream_ptr(dy)), "conv_wgrad")


_patch_ws = {}          # device -> scratch of clite_conv_wgrad_patch_workspace() bytes (the kernel's per-workgroup partial sums: no state between calls)


def conv_wgrad_patch_applies(cv):
    """The problems clite_conv_wgrad_patch covers by shape (bf16, 3 x 3 / stride 1 / pad 1, 64 -> 64, rows of at most 58 pixels); the library
    still declines (returns 1) in the deterministic mode or under a forced tile policy."""
    return cv.dtype == BF16 and cv.C == 64 and cv.K == 64 and cv.R == 3 and cv.S == 3 and cv.stride == 1 and cv.pad == 1 and cv.W <= 58


def _indexed(device):
    """torch.device("cuda") and torch.device("cuda", 0) are different dictionary keys: always the indexed form (a tensor's .device carries the index)."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


def patch_workspace(device):
    """The device's scratch for clite_conv_wgrad_patch (37.7 MB). DeviceRuntime creates it when the model moves to the GPU, i.e. before any
    stream capture; a first use INSIDE a capture would put it into the graph's private pool."""
    device = _indexed(device)
    ws = _patch_ws.get(device)
    if ws is None:
        if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("clip_lite_amd: the patch-resident weight gradient's workspace must exist before a stream capture (hip.patch_workspace)")
        n = C.c_uint64(0)
        check(lib().clite_conv_wgrad_patch_workspace(C.byref(n)), "conv_wgrad_patch_workspace")
        ws = _patch_ws[device] = torch.empty(n.value, dtype=torch.uint8, device=device)
    return ws


def conv_wgrad_patch(dy, x, cv, dw):
    """dw += the weight gradient on the patch-resident kernel; falls back to clite_conv_wgrad when the library declines. Launches on the current
    stream. Every call on one device shares one scratch buffer: the members of a backward pass that qualify (ResNet layer1's 3 x 3 convolutions)
    all belong to ONE weight-gradient group, i.e. one stream, which orders them."""
    ws = patch_workspace(dy.device)
    rc = lib().clite_conv_wgrad_patch(p(dy), p(x), C.byref(cv), p(dw), p(ws), ws.numel(), stream_ptr(dy))
    if rc == 1:
        conv_wgrad(dy, x, cv, dw)
    else:
        check(rc, "conv_wgrad_patch")


import collections
_inflight_groups = collections.deque()
_ws_pool = []          # [(event recorded behind the last launch that used it, device workspace, pinned host workspace)] of uncaptured WgradGroup launches


KCHUNK_TILES = 256          # K tiles (of 32) per workgroup of a grouped weight gradient: csrc/gemm_group.hip KCHUNK


class WgradGroup:
    """Weight gradients collected during a backward pass and launched together (include/clite.h: clite_wgrad_group). The backward executors
    call conv() / linear() where they would have launched clite_conv_wgrad / clite_gemm_tn, and call() for anything else that only feeds
    the gradient arena (bias column sums, the stem's packed gradient); launch() enqueues the lot on the current stream. The object keeps
    every operand referenced and owns the descriptor workspace (device + pinned host), so it must outlive a captured graph that contains
    its launch."""

    wide = True          # class-wide switch (tools/ab_runtime.py hip.WgradGroup.wide=0): False keeps every member on the 4-wave tiles (CLITE_WGRAD_NARROW)

    def __init__(self, dt, workspace=None, zeroed=False):
        """workspace: optional pre-allocated (device uint8 tensor, pinned host uint8 tensor) of equal size — required when launch() runs
        inside a stream capture, where pinned memory cannot be allocated (alloc_workspace()). zeroed: the caller vouches that every member's
        gradient buffer holds zeros when the group is launched (CLITE_WGRAD_ZEROED: plain stores instead of float atomics for the members whose
        contraction fits one K chunk) — the captured train step, which visits every weight once per step behind an update that zeroes the arena."""
        self.dt, self.items, self.extra, self.keep, self.wgs = dt, [], [], [], 0
        self.post = []          # after(): launches that read what the grouped launch wrote - right behind it, on ITS stream (never with the extras)
        self.kflags = 0x200 if zeroed else 0
        self.ws_dev, self.ws_host = workspace if workspace is not None else (None, None)

    @staticmethod
    def alloc_workspace(device, nbytes=4 << 20):
        """4 MiB holds the descriptors of ~30 000 workgroups under the library's worst-case list padding (clite_wgrad_group_workspace: 8 x 16 B
        per workgroup); ResNet-101 + BERT-base at batch 256 need about 8 000."""
        return torch.empty(nbytes, dtype=torch.uint8, device=device), torch.empty(nbytes, dtype=torch.uint8).pin_memory()

    @staticmethod
    def _wgs(M, N, K, bm=128, bn=128):
        kt = (K + 31) // 32
        return ((M + bm - 1) // bm) * ((N + bn - 1) // bn) * ((kt + KCHUNK_TILES - 1) // KCHUNK_TILES)

    # class-wide switch: True sends the 64 -> 64 3 x 3 members to the patch-resident kernel (clite_conv_wgrad_patch) instead of the grouped launch.
    # OFF: stand-alone the kernel takes 45 + 8 us per member where the member alone in a group takes 97 — but INSIDE layer1's group the three
    # members cost ~25 us each of marginal time (they fill slots the HBM-bound 1 x 1 members leave idle), and three launches behind the group were
    # measured 0.1 ms slower per step (same box, tools/ab_runtime.py hip.WgradGroup.patch=1: 15.34 / 15.46 vs 15.22 / 15.36 ms). Kept for the
    # ungrouped paths (hip.conv_wgrad_patch) and as a tested entry point of the C ABI.
    patch = False

    def conv(self, dy, x, cv, dw, row_scale=None, short_k=False):
        """row_scale: f32 [K] per-output-channel factor on this member's product (clite_wgrad_item.row_scale; the folded BatchNorm backward).
        short_k: K chunks of a quarter of the usual length (CLITE_WGRAD_SHORTK: a tiny output over a very long contraction)."""
        if row_scale is None and self.patch and self.wide and conv_wgrad_patch_applies(cv) and not is_deterministic():
            # the 64 -> 64 3 x 3 members run on the patch-resident kernel behind the grouped launch (97 -> ~30 us each): a launch of their own,
            # on the stream of launch()
            self.keep += [dy, x, dw]
            self.extra.append(lambda: conv_wgrad_patch(dy, x, cv, dw))
            return
        it = WgradItem()
        it.kind, it.a, it.b, it.out, it.cv = (0 if self.wide else 0x100) | self.kflags | (0x400 if short_k else 0), p(dy), p(x), p(dw), cv
        it.row_scale = p(row_scale)
        self.items.append(it)
        self.keep += [dy, x, dw, row_scale]
        ncols = cv.R * cv.S * cv.C
        self.wgs += self._wgs(cv.K, ncols, cv.N * cv.Ho * cv.Wo, 64 if cv.K <= 64 else 128, 64 if ncols <= 64 else 128) * (4 if short_k else 1)

    def conv_fp8(self, dy8, x8, cv, dw):
        """The same weight gradient on fp8 operands (clite_wgrad_item kind 2, BASELINE configs[4]): dy8 = Fp8View of the e5m2 gradient copy that
        bn_bwd_apply's fused quantiser wrote, x8 = Fp8View of the e4m3 activation copy that bn_apply wrote; the 256 x 256 tile on the block-scaled MFMA."""
        it = WgradItem()
        it.kind, it.a, it.b, it.out, it.cv = 2 | self.kflags, p(dy8.q), p(x8.q), p(dw), cv
        it.a_scales, it.b_scales = p(dy8.scales), p(x8.scales)
        self.items.append(it)
        self.keep += [dy8.q, x8.q, dy8.scales, x8.scales, dw]
        self.wgs += self._wgs(cv.K, cv.R * cv.S * cv.C, cv.N * cv.Ho * cv.Wo, 256, 256)

    def linear(self, A, B, M, N, K, out, lda=None, ldb=None, ldc=None):
        """out[M][N] += A[K][M]^T B[K][N] (f32 accumulate)."""
        it = WgradItem()
        it.kind, it.a, it.b, it.out = (1 if self.wide else 0x101) | self.kflags, p(A), p(B), p(out)
        it.M, it.N, it.K, it.lda, it.ldb, it.ldc = M, N, K, lda or M, ldb or N, ldc or N
        self.items.append(it)
        self.keep += [A, B, out]
        self.wgs += self._wgs(M, N, K)

    def call(self, fn):
        self.extra.append(fn)

    def after(self, fn):
        """fn() launches something that READS a member's result (the folded BatchNorm backward's correction terms: they need the group's Gram matrix):
        enqueued by launch() right behind the grouped launch, on the same stream - unlike the extras, which a caller may replay elsewhere."""
        self.post.append(fn)

    def launch_extras(self):
        """The members that are launches of their own (bias column sums, the stem's packed gradient, the patch-resident 3 x 3 weight gradients):
        on the current stream. The captured step runs the last segment's on the text encoder's stream, beside the grouped launch."""
        for fn in self.extra:
            fn()

    def launch(self, extras=True):
        if self.items:
            arr = (WgradItem * len(self.items))(*self.items)
            first = self.keep[0]
            pooled = None
            if self.ws_dev is None and first.is_cuda:
                nbytes = C.c_uint64(0)
                check(lib().clite_wgrad_group_workspace(len(self.items), self.wgs, C.byref(nbytes)), "wgrad_group_workspace")
                # uncaptured launch: a workspace from the pool of finished ones (allocating pinned memory costs ~8 ms per call: measured as 15 ms
                # of a 59 ms eager step). A pool entry is reusable once the event recorded behind the launch that used it has completed.
                for i, (ev_, dev_, host_) in enumerate(_ws_pool):
                    if dev_.numel() >= nbytes.value and dev_.device == first.device and ev_.query():
                        pooled = _ws_pool.pop(i)
                        break
                if pooled is None:
                    n_alloc = max(int(nbytes.value), 4 << 20)
                    pooled = (None, torch.empty(n_alloc, dtype=torch.uint8, device=first.device), torch.empty(n_alloc, dtype=torch.uint8).pin_memory())
                self.ws_dev, self.ws_host = pooled[1], pooled[2]
            nb = self.ws_dev.numel() if self.ws_dev is not None else 0
            check(lib().clite_wgrad_group(self.dt, arr, len(self.items), p(self.ws_dev), self.ws_host.data_ptr() if self.ws_host is not None else None,
                                          nb, stream_ptr(first)), "wgrad_group")
            if first.is_cuda and not torch.cuda.is_current_stream_capturing():
                # the descriptor copy and the launches are only enqueued: keep this object (its pinned staging, its operands) alive until the
                # stream has passed them — the host allocator knows nothing about the library's own hipMemcpyAsync
                ev = torch.cuda.Event()
                ev.record()
                if pooled is not None:
                    _ws_pool.append((ev, pooled[1], pooled[2]))
                    self.ws_dev = self.ws_host = None
                while _inflight_groups and _inflight_groups[0][0].query():
                    _inflight_groups.popleft()
                _inflight_groups.append((ev, self))
        for fn in self.post:
            fn()
        if extras:
            self.launch_extras()


class Fp8Tensor:
    """An e4m3 copy of a tensor with its per-tensor scales (clite_fp8_quantize): q (uint8, same shape), scales = device f32 {scale, 1/scale}."""
    __slots__ = ("q", "scales", "amax")

    def __init__(self, x, dt):
        self.q = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        self.amax = torch.zeros(1, dtype=torch.float32, device=x.device)
        self.scales = torch.empty(2, dtype=torch.float32, device=x.device)
        check(lib().clite_fp8_quantize(dt, p(x), x.numel(), p(self.amax), p(self.scales), p(self.q), stream_ptr(x)), "fp8_quantize")


class Fp8View:
    """An e4m3 tensor somebody else produced (bn_apply's fused copy, a slice of the grouped weight quantiser's arena): q + its scales."""
    __slots__ = ("q", "scales")

    def __init__(self, q, scales):
        self.q, self.scales = q, scales


FP8_GROUP_CHUNK = 8192


class Fp8WeightGroup:
    """Per-tensor current-scaling e4m3 copies of many tensors of ONE bf16 arena in two launches (clite_fp8_quantize_group): the conv weights
    after every optimizer update. `spans` = [(element offset, numel)]; view(i, shape) is tensor i's Fp8View."""

    def __init__(self, arena_lp, spans):
        dev = arena_lp.device
        self.base, self.n = arena_lp, len(spans)
        items = (Fp8Item * self.n)()
        table = []
        lo, hi = min(o for o, _ in spans), max(o + n for o, n in spans)
        self.lo = lo
        for i, (o, n) in enumerate(spans):
            assert o % 8 == 0 and n % 8 == 0 and n <= 4096 * FP8_GROUP_CHUNK and i < (1 << 20)
            items[i].offset, items[i].numel = o, n
            table += [(i << 12) | c for c in range((n + FP8_GROUP_CHUNK - 1) // FP8_GROUP_CHUNK)]
        self.spans = list(spans)
        self.items = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(dev)
        assert max(table) < 2 ** 31
        self.table = torch.tensor(table, dtype=torch.int32).to(dev)
        self.n_wgs = len(table)
        self.q = torch.zeros(hi, dtype=torch.uint8, device=dev)          # indexed by the arena's element offsets (the part below `lo` is never touched)
        self.amax = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.partial = torch.zeros(self.n_wgs, dtype=torch.int32, device=dev)
        self.scales = torch.ones(self.n, 2, dtype=torch.float32, device=dev)

    def quantize(self):
        check(lib().clite_fp8_quantize_group(p(self.base), p(self.items), p(self.table), self.n, self.n_wgs, p(self.partial), p(self.amax), p(self.scales), p(self.q),
                                             stream_ptr(self.base)), "fp8_quantize_group")

    def view(self, i, shape):
        o, n = self.spans[i]
        return Fp8View(self.q[o:o + n].view(shape), self.scales[i])


FP8_AMAX_WORDS = 16 * 32          # one amax slot (include/clite.h: CLITE_FP8_AMAX_REPLICAS x CLITE_FP8_AMAX_STRIDE floats)


def fp8_scale_update(amax, scales):
    """amax: f32 [n][FP8_AMAX_WORDS] slots, scales: f32 [n][2]."""
    assert amax.numel() % FP8_AMAX_WORDS == 0 and scales.numel() == 2 * (amax.numel() // FP8_AMAX_WORDS)
    check(lib().clite_fp8_scale_update(p(amax), p(scales), amax.numel() // FP8_AMAX_WORDS, stream_ptr(amax)), "fp8_scale_update")


def gemm_nt_fp8(A8, B8, M, N, K, ep, lda=None, ldb=None):
    check(lib().clite_gemm_nt_fp8(p(A8.q), lda or K, p(B8.q), ldb or K, M, N, K, p(A8.scales), p(B8.scales), C.byref(ep), stream_ptr(A8.q)), "gemm_nt_fp8")


def conv_fwd_fp8(x8, w8, cv, ep):
    check(lib().clite_conv_fwd_fp8(p(x8.q), p(w8.q), C.byref(cv), p(x8.scales), p(w8.scales), C.byref(ep), stream_ptr(x8.q)), "conv_fwd_fp8")


def conv_dgrad_fp8(dy8, wt8, cv, ep):
    """dy8: the e5m2 copy of dy (bn_bwd_apply's fused quantiser), wt8: the e4m3 copy of the transposed weights [C][R][S][K]; ep: the BatchNorm-backward
    form only (relu_bits + bn + colsum). include/clite.h: clite_conv_dgrad_fp8."""
    check(lib().clite_conv_dgrad_fp8(p(dy8.q), p(wt8.q), C.byref(cv), p(dy8.scales), p(wt8.scales), C.byref(ep), stream_ptr(dy8.q)), "conv_dgrad_fp8")


def stem_fwd(dt, xpad, wv, N, Hp, Wp, Ho, Wo, ep):
    check(lib().clite_stem_fwd(p(xpad), p(wv), dt, N, Hp, Wp, Ho, Wo, C.byref(ep), stream_ptr(xpad)), "stem_fwd")


def stem_wgrad(dt, dy, xpad, N, Hp, Wp, Ho, Wo, dwv):
    check(lib().clite_stem_wgrad(p(dy), p(xpad), dt, N, Hp, Wp, Ho, Wo, p(dwv), stream_ptr(dy)), "stem_wgrad")


def stem_wgrad_patch(dt, dy, xpad, N, Hp, Wp, Ho, Wo, dw):
    """The stem's weight gradient on the patch-resident kernel, += into the f32 [64][7][7][3] gradient. False: not covered (the caller takes
    stem_wgrad + stem_unpack_grad). The device's patch workspace must exist (patch_workspace: DeviceRuntime creates it)."""
    ws = _patch_ws.get(_indexed(dy.device))
    if ws is None:
        return False
    rc = lib().clite_stem_wgrad_patch(p(dy), p(xpad), dt, N, Hp, Wp, Ho, Wo, p(dw), p(ws), ws.numel(), stream_ptr(dy))
    if rc == 1:
        return False
    check(rc, "stem_wgrad_patch")
    return True


def stem_pack(dt, w, wv):
    check(lib().clite_stem_pack(p(w), p(wv), dt, stream_ptr(w)), "stem_pack")


def stem_unpack_grad(dwv, dw):
    check(lib().clite_stem_unpack_grad(p(dwv), p(dw), stream_ptr(dwv)), "stem_unpack_grad")


# ------------------------------------------------------------------------------------------------ BN / pools
def bn_desc(M, Cc, stats, gamma, beta, rmean, rvar, training, update, momentum, eps, relu, res_bn=None, centered=False, relu_bits=None, fp8=None, out_sum=None):
    """stats: hip.Stats (training) or None (eval: running statistics). relu_bits: uint8 [M][C / 8] that bn_apply fills with the packed ReLU mask.
    fp8: (q uint8 [M][C] or None, scales f32[2] or None, amax f32[1] or None) — bn_apply's producer-fused e4m3 copy (clite_bn.fp8_*)."""
    b = Bn()
    b.M, b.C = M, Cc
    b.centered = int(centered)
    b.replicas, b.rstride = (stats.R, stats.rstride) if stats is not None else (1, 0)
    b.stats = p(stats.t) if stats is not None else None
    b.gamma, b.beta, b.running_mean, b.running_var = p(gamma), p(beta), p(rmean), p(rvar)
    b.training, b.update_running, b.momentum, b.eps, b.relu = int(training), int(update), momentum, eps, int(relu)
    if res_bn is not None:
        rs, rg, rb, rrm, rrv = res_bn
        b.res_stats = p(rs.t) if rs is not None else None
        b.res_gamma, b.res_beta, b.res_running_mean, b.res_running_var = p(rg), p(rb), p(rrm), p(rrv)
    b.relu_bits = p(relu_bits)
    if fp8 is not None:
        b.fp8_out, b.fp8_scale, b.fp8_amax = p(fp8[0]), p(fp8[1]), p(fp8[2])
    if out_sum is not None:          # Stats: row 0 of every replica receives the column sums of the stored output (clite_bn.out_sum)
        b.out_sum, b.out_sum_replicas, b.out_sum_stride = p(out_sum.t), out_sum.R, out_sum.rstride
    return b


def bn_centered_var(dt, y, stats, M, Cc):
    check(lib().clite_bn_centered_var(dt, p(y), p(stats.t), stats.R, stats.rstride, M, Cc, stream_ptr(y)), "bn_centered_var")


def bn_apply(dt, bn, y, res, out):
    check(lib().clite_bn_apply(C.byref(bn), dt, p(y), p(res), p(out), stream_ptr(y)), "bn_apply")


def _mask_args(mask):
    """A ReLU mask is either a tensor of the compute dtype (sign test) or packed bits (uint8, as bn_apply's relu_bits writes them)."""
    if mask is not None and mask.dtype == torch.uint8:
        return None, p(mask)
    return p(mask), None


def bn_bwd_reduce(dt, dout, mask, y, stats, dstats, M, Cc):
    assert stats.R == dstats.R
    mt, mb = _mask_args(mask)
    check(lib().clite_bn_bwd_reduce(dt, p(dout), mt, mb, p(y), p(stats.t), p(dstats.t), stats.R, stats.rstride, M, Cc, stream_ptr(y)), "bn_bwd_reduce")


def bn_bwd_apply(dt, bn, dout, mask, y, dstats, dy, dz, dgamma, dbeta):
    mt, mb = _mask_args(mask)
    check(lib().clite_bn_bwd_apply(C.byref(bn), dt, p(dout), mt, mb, p(y), p(dstats.t), p(dy), p(dz), p(dgamma), p(dbeta), stream_ptr(y)),
          "bn_bwd_apply")


def maxpool_fwd(dt, x, out, idx, N, H, W, Cc):
    check(lib().clite_maxpool3x3s2_fwd(dt, p(x), p(out), p(idx), N, H, W, Cc, stream_ptr(x)), "maxpool_fwd")


def maxpool_bwd(dt, dout, idx, dx, N, H, W, Cc):
    check(lib().clite_maxpool3x3s2_bwd(dt, p(dout), p(idx), p(dx), N, H, W, Cc, stream_ptr(dout)), "maxpool_bwd")


def stem_bn_pool_fwd(dt, bn, y, pooled, idx, N, H, W, ymax=None):
    """ymax (and bn.relu_bits): the pooled-size operands of the backward reductions (include/clite.h: clite_stem_bn_pool_fwd_ex)."""
    check(lib().clite_stem_bn_pool_fwd_ex(C.byref(bn), dt, p(y), p(pooled), p(idx), p(ymax), N, H, W, stream_ptr(y)), "stem_bn_pool_fwd")


def stem_bn_pool_bwd_apply(dt, bn, dpool, idx, y, dstats, dy, dgamma, dbeta, N, H, W):
    check(lib().clite_stem_bn_pool_bwd_apply(C.byref(bn), dt, p(dpool), p(idx), p(y), p(dstats.t), p(dy), p(dgamma), p(dbeta), N, H, W, stream_ptr(y)),
          "stem_bn_pool_bwd_apply")


def stem_bn_pool_bwd(dt, bn, dpool, idx, y, dstats, dy, dgamma, dbeta, N, H, W):
    check(lib().clite_stem_bn_pool_bwd(C.byref(bn), dt, p(dpool), p(idx), p(y), p(dstats.t), p(dy), p(dgamma), p(dbeta), N, H, W, stream_ptr(y)), "stem_bn_pool_bwd")


def avgpool_fwd(dt, x, out, N, HW, Cc):
    check(lib().clite_avgpool_fwd(dt, p(x), p(out), N, HW, Cc, stream_ptr(x)), "avgpool_fwd")


def avgpool_bwd(dt, dout, dx, N, HW, Cc):
    check(lib().clite_avgpool_bwd(dt, p(dout), p(dx), N, HW, Cc, stream_ptr(dout)), "avgpool_bwd")


def image_to_nhwc4(dt, img, out, N, H, W, pad, Hp, Wp):
    check(lib().clite_image_to_nhwc4(dt, p(img), p(out), N, H, W, pad, Hp, Wp, stream_ptr(img)), "image_to_nhwc4")


def colsum(dt, x, out, M, N):
    check(lib().clite_colsum(dt, p(x), p(out), M, N, stream_ptr(x)), "colsum")


# ------------------------------------------------------------------------------------------------ BERT pieces
NO_DROP = (0.0, 0, 0)


def layernorm_fwd(dt, x, gamma, beta, eps, out, stats, M, Cc, drop=NO_DROP, fp8=None):
    """fp8 = (q | None, scales | None, amax slot | None): also leave the e4m3 copy of `out` / record max |out| (clite_layernorm_fwd_q8, bf16)."""
    if fp8 is not None:
        check(lib().clite_layernorm_fwd_q8(dt, p(x), p(gamma), p(beta), eps, p(out), p(stats), M, Cc, drop[0], drop[1], drop[2],
                                           p(fp8[0]), p(fp8[1]), p(fp8[2]), stream_ptr(x)), "layernorm_fwd_q8")
        return
    check(lib().clite_layernorm_fwd(dt, p(x), p(gamma), p(beta), eps, p(out), p(stats), M, Cc, drop[0], drop[1], drop[2], stream_ptr(x)),
          "layernorm_fwd")


def layernorm_bwd(dt, dy, x, stats, gamma, dx, dx_masked, dgamma, dbeta, M, Cc, drop_in=NO_DROP, drop_out=NO_DROP, dcolsum=None):
    check(lib().clite_layernorm_bwd(dt, p(dy), p(x), p(stats), p(gamma), p(dx), p(dx_masked), p(dgamma), p(dbeta), p(dcolsum), M, Cc,
                                    drop_in[0], drop_in[1], drop_in[2], drop_out[0], drop_out[1], drop_out[2], stream_ptr(x)), "layernorm_bwd")


def embed_fwd(dt, ids, word, pos, typ, out, M, L, Cc, vocab):
    check(lib().clite_embed_fwd(dt, p(ids), p(word), p(pos), p(typ), p(out), M, L, Cc, vocab, stream_ptr(out)), "embed_fwd")


def embed_bwd(dt, ids, d, dword, dpos, M, L, Cc, vocab, padding_idx=-1):
    check(lib().clite_embed_bwd(dt, p(ids), p(d), p(dword), p(dpos), M, L, Cc, vocab, padding_idx, stream_ptr(d)), "embed_bwd")


def attention_fwd(dt, qkv, mask, ctx, B, L, H, drop=NO_DROP):
    check(lib().clite_attention_fwd(dt, p(qkv), p(mask), p(ctx), B, L, H, drop[0], drop[1], drop[2], stream_ptr(qkv)), "attention_fwd")


def attention_bwd(dt, qkv, mask, dctx, dqkv, B, L, H, drop=NO_DROP):
    check(lib().clite_attention_bwd(dt, p(qkv), p(mask), p(dctx), p(dqkv), B, L, H, drop[0], drop[1], drop[2], stream_ptr(qkv)), "attention_bwd")


def tanh_bwd(dt, dy, y, out, n):
    check(lib().clite_tanh_bwd(dt, p(dy), p(y), p(out), n, stream_ptr(dy)), "tanh_bwd")


# ------------------------------------------------------------------------------------------------ loss / update
def critic_jsd_fwd(dt, f1, f2, temperature, B, D, work, acc, neg=None):
    check(lib().clite_critic_jsd_fwd(dt, p(f1), p(f2), p(temperature), B, D, p(neg), p(work), p(acc), stream_ptr(f1)), "critic_jsd_fwd")


def l2_normalize(dt, x, out, B, D):
    check(lib().clite_l2_normalize(dt, p(x), p(out), B, D, stream_ptr(x)), "l2_normalize")


def l2_normalize_bwd(dt, x, y, dy, dx, B, D):
    check(lib().clite_l2_normalize_bwd(dt, p(x), p(y), p(dy), p(dx), B, D, stream_ptr(x)), "l2_normalize_bwd")


def infonce_fwd(Cm, ld, B, temperature, lse_r, lse_c, acc):
    check(lib().clite_infonce_fwd(p(Cm), ld, B, p(temperature), p(lse_r), p(lse_c), p(acc), stream_ptr(Cm)), "infonce_fwd")


def infonce_bwd(dt, Cm, ld, B, temperature, lse_r, lse_c, gout, scale, dC, ldd, dtemp):
    check(lib().clite_infonce_bwd(dt, p(Cm), ld, B, p(temperature), p(lse_r), p(lse_c), p(gout), scale, p(dC), ldd, p(dtemp), stream_ptr(Cm)), "infonce_bwd")


def critic_jsd_bwd(dt, f1, f2, temperature, work, gout, scale, B, D, df1, df2, dtemp, neg=None, neg_inv=None):
    check(lib().clite_critic_jsd_bwd(dt, p(f1), p(f2), p(temperature), p(work), p(gout), scale, B, D, p(neg), p(neg_inv), p(df1), p(df2), p(dtemp),
                                     stream_ptr(f1)),
          "critic_jsd_bwd")


def prior_tail_fwd(dt, h1, w2, b2, B, K, logit, acc, softplus=False):
    check(lib().clite_prior_tail_fwd(dt, p(h1), p(w2), p(b2), B, K, int(softplus), p(logit), p(acc), stream_ptr(h1)), "prior_tail_fwd")


def prior_tail_bwd(dt, h1, w2, logit, gout, scale, B, K, dh1, dw2, db2):
    check(lib().clite_prior_tail_bwd(dt, p(h1), p(w2), p(logit), p(gout), scale, B, K, p(dh1), p(dw2), p(db2), stream_ptr(h1)), "prior_tail_bwd")


def add(dt, a, b, out):
    check(lib().clite_add(dt, p(a), p(b), p(out), a.numel(), stream_ptr(a)), "add")


def loss_finalize(acc, prior_weight, out):
    check(lib().clite_loss_finalize(p(acc), prior_weight, p(out), stream_ptr(acc)), "loss_finalize")


def uniform_fill(dt, out, n, seed, site):
    check(lib().clite_uniform_fill(dt, p(out), n, seed, site, stream_ptr(out)), "uniform_fill")


def sumsq(x, n, out, partials):
    check(lib().clite_sumsq(p(x), n, p(out), p(partials), partials.numel(), stream_ptr(x)), "sumsq")


def sum_slices(src, slices, stride, n, dst):
    check(lib().clite_sum_slices(p(src), slices, stride, n, p(dst), stream_ptr(src)), "sum_slices")


def sgd_step(pf, gf, vf, slow, cast, items_ptr, n_items, hp, ss):
    check(lib().clite_sgd_step(p(pf), p(gf), p(vf), p(slow), p(cast), items_ptr, n_items, p(hp), p(ss), stream_ptr(pf)), "sgd_step")


def adamw_step(pf, gf, mf, v2f, slow, cast, items_ptr, n_items, hp, ss):
    check(lib().clite_adamw_step(p(pf), p(gf), p(mf), p(v2f), p(slow), p(cast), items_ptr, n_items, p(hp), p(ss), stream_ptr(pf)), "adamw_step")


def cast_bf16(src, dst, n):
    check(lib().clite_cast_bf16(p(src), p(dst), n, stream_ptr(src)), "cast_bf16")
