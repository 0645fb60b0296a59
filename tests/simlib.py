"""ctypes access to the wave-simulator build of the kernel sources (tests only).

`tests/wavesim/_build/libclite_sim.so` is the SAME csrc/*.hip code compiled for the host with the lane-accurate
emulation in tests/wavesim/wavesim.h. It exists so kernel index math can be checked on a box with no GPU; it is
never imported by the product package (clip-lite_amd/), which only loads the gfx950 library.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM_SO = os.path.join(ROOT, "tests", "wavesim", "_build", "libclite_sim.so")


class Epilogue(C.Structure):
    _fields_ = [
        ("out", C.c_void_p), ("ldc", C.c_int32), ("out_f32", C.c_int32), ("atomic", C.c_int32),
        ("alpha", C.c_float), ("bias", C.c_void_p), ("act", C.c_int32), ("preact", C.c_void_p),
        ("dact_aux", C.c_void_p), ("dact", C.c_int32), ("drop_p", C.c_float), ("drop_seed", C.c_uint64),
        ("drop_site", C.c_uint32), ("residual", C.c_void_p), ("colsum", C.c_void_p),
        ("colsum_replicas", C.c_int32), ("colsum_stride", C.c_int32), ("colsum_rows", C.c_int32),
        ("bn_y", C.c_void_p), ("bn_stats", C.c_void_p), ("bn_replicas", C.c_int32), ("bn_rstride", C.c_int32),
        ("bn_inv_count", C.c_float), ("mask_after_residual", C.c_int32), ("relu_bits", C.c_void_p), ("splitk_ws", C.c_void_p),
        ("residual_subsample", C.c_int32), ("fp8_out", C.c_void_p), ("fp8_scale", C.c_void_p), ("fp8_amax", C.c_void_p),
    ]


class Conv(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dtype", "N", "H", "W", "C", "K", "R", "S", "stride", "pad", "Ho", "Wo")]


def build_sim():
    subprocess.check_call(["make", "-s", "sim"], cwd=ROOT)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_sim()
        _lib = C.CDLL(SIM_SO)
    return _lib


def to_bf16(x):
    """float32 ndarray -> uint16 bf16 bits, round-to-nearest-even."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return r.astype(np.uint16)


def from_bf16(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def bf16_round(x):
    return from_bf16(to_bf16(x))


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_ep(out, ldc, out_f32=False, atomic=False, alpha=1.0, bias=None, act=0, preact=None, dact_aux=None,
            dact=0, drop_p=0.0, drop_seed=0, drop_site=0, residual=None, colsum=None, relu_bits=None, fp8=None):
    ep = Epilogue()
    ep.out = ptr(out); ep.ldc = ldc; ep.out_f32 = int(out_f32); ep.atomic = int(atomic); ep.alpha = alpha
    ep.bias = ptr(bias); ep.act = act; ep.preact = ptr(preact); ep.dact_aux = ptr(dact_aux); ep.dact = dact
    ep.drop_p = drop_p; ep.drop_seed = drop_seed; ep.drop_site = drop_site
    ep.residual = ptr(residual); ep.colsum = ptr(colsum); ep.relu_bits = ptr(relu_bits)
    if fp8 is not None:
        ep.fp8_out, ep.fp8_scale, ep.fp8_amax = ptr(fp8[0]), ptr(fp8[1]), ptr(fp8[2])
    return ep


def pack_relu_bits(a):
    """[M][C] activations -> uint8 [M][C / 8]: bit e of byte (m, c / 8) = a[m][c + e] > 0 (clite_bn.relu_bits layout)."""
    return np.packbits(np.asarray(a) > 0, axis=-1, bitorder="little")


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


def prep(x, dtype):
    """(values the kernel sees as f32, buffer to pass) for dtype 0 = bf16, 1 = f32"""
    if dtype == 0:
        xr = bf16_round(x)
        return xr, to_bf16(xr)
    x = np.ascontiguousarray(x, np.float32)
    return x, x


def outbuf(shape, dtype):
    return np.zeros(shape, np.uint16 if dtype == 0 else np.float32)


def val(buf, dtype):
    return from_bf16(buf) if dtype == 0 else buf
