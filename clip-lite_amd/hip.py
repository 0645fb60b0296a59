"""ctypes binding of libclite_hip.so (C ABI: include/clite.h) — the only route from the Python host layer to device math.

There is deliberately no fallback: if the gfx950 library is missing or a kernel returns an error, the caller gets a
RuntimeError. PyTorch is used for device memory, streams and torch.distributed only.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libclite_hip.so")
ABI_VERSION = 1

ACT_NONE, ACT_RELU, ACT_GELU, ACT_TANH = 0, 1, 2, 3


class Epilogue(C.Structure):
    _fields_ = [
        ("out", C.c_void_p), ("ldc", C.c_int32), ("out_f32", C.c_int32), ("atomic", C.c_int32),
        ("alpha", C.c_float), ("bias", C.c_void_p), ("act", C.c_int32), ("preact", C.c_void_p),
        ("dact_aux", C.c_void_p), ("dact", C.c_int32), ("drop_p", C.c_float), ("drop_seed", C.c_uint64),
        ("drop_site", C.c_uint32), ("residual", C.c_void_p), ("colsum", C.c_void_p),
    ]


class Conv(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dtype", "N", "H", "W", "C", "K", "R", "S", "stride", "pad", "Ho", "Wo")]


_lib = None


def lib():
    """Load the kernel library once; fail loudly when it is absent (no CPU or eager fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"clip_lite_amd: {LIB_PATH} not found. Build it with `make hip` (or __graft_entry__.build()); "
                "this package has no fallback path.")
        l = C.CDLL(LIB_PATH)
        l.clite_abi_version.restype = C.c_int
        if l.clite_abi_version() != ABI_VERSION:
            raise RuntimeError("clip_lite_amd: libclite_hip.so ABI version mismatch; rebuild with `make hip`")
        _lib = l
    return _lib


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def p(t):
    """Device pointer of a tensor (None -> NULL). The tensor must be contiguous in the layout the kernel expects."""
    return None if t is None else C.c_void_p(t.data_ptr())


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"clip_lite_amd: {what} failed with code {rc}")


def epilogue(out, ldc, atomic=False, alpha=1.0, bias=None, act=ACT_NONE, preact=None, dact_aux=None, dact=0,
             drop_p=0.0, drop_seed=0, drop_site=0, residual=None, colsum=None):
    ep = Epilogue()
    ep.out = out.data_ptr(); ep.ldc = ldc
    ep.out_f32 = int(out.dtype == torch.float32)
    assert out.dtype in (torch.float32, torch.bfloat16)
    ep.atomic = int(atomic); ep.alpha = alpha
    ep.bias = None if bias is None else bias.data_ptr()
    ep.act = act
    ep.preact = None if preact is None else preact.data_ptr()
    ep.dact_aux = None if dact_aux is None else dact_aux.data_ptr()
    ep.dact = dact
    ep.drop_p = drop_p; ep.drop_seed = drop_seed; ep.drop_site = drop_site
    ep.residual = None if residual is None else residual.data_ptr()
    ep.colsum = None if colsum is None else colsum.data_ptr()
    return ep
