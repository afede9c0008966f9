"""ctypes binding of libstes_hip.so (the C-ABI declared in include/stes.h).

The library is the product: if it is missing or a kernel launch fails this module
raises, it never falls back to another implementation.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libstes_hip.so")

_lib = None


class StesError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise StesError(f"{LIB_PATH} not found: build it with `python -m spot_amd.build` "
                            "(hipcc --offload-arch=gfx950); there is no fallback path")
        _lib = C.CDLL(LIB_PATH)
        _lib.stes_last_error.restype = C.c_char_p
        _lib.stes_version.restype = C.c_int
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise StesError(f"{what} failed ({rc}): {lib().stes_last_error().decode()}")


def stream_ptr() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (or None)."""
    if t is None:
        return None
    return t.data_ptr()


_P = C.c_void_p
_I32 = C.c_int32
_I64 = C.c_int64
_F32 = C.c_float


def _struct(name, fields):
    return type(name, (C.Structure,), {"_fields_": fields})


GemmArgs = _struct("GemmArgs", [
    ("A", _P), ("lda", _I64), ("A0", _P), ("lda0", _I64), ("k0", _I32),
    ("a_scale", _P), ("a_shift", _P), ("a_relu", _I32),
    ("a_gate", _P), ("gate_group", _I32),
    ("map_stride", _I32), ("Ho", _I32), ("Wo", _I32), ("Hi", _I32), ("Wi", _I32),
    ("W", _P), ("ldw", _I64), ("bias", _P), ("act", _I32),
    ("C", _P), ("ldc", _I64), ("c_f32", _I32),
    ("col_sum", _P), ("col_sumsq", _P),
    ("M", _I64), ("N", _I32), ("K", _I32)])

WgradArgs = _struct("WgradArgs", [
    ("P", _P), ("ldp", _I64), ("Q", _P), ("ldq", _I64), ("Q0", _P), ("ldq0", _I64), ("k0", _I32),
    ("q_scale", _P), ("q_shift", _P), ("q_relu", _I32), ("q_gate", _P), ("gate_group", _I32),
    ("map_stride", _I32), ("Ho", _I32), ("Wo", _I32), ("Hi", _I32), ("Wi", _I32),
    ("dW", _P), ("lddw", _I64), ("dbias", _P),
    ("M", _I64), ("N1", _I32), ("N2", _I32)])


def gemm(A, W, Cout, M, N, K, *, lda=None, ldw=None, ldc=None, A0=None, lda0=0, k0=0,
         scale=None, shift=None, relu=False, gate=None, gate_group=0, rowmap=None,
         bias=None, act=0, stats=None):
    """C = act(pro(A) W^T + bias); see stes_gemm in include/stes.h."""
    a = GemmArgs()
    a.A, a.lda = ptr(A), lda if lda is not None else K
    a.A0, a.lda0, a.k0 = ptr(A0), lda0, k0
    a.a_scale, a.a_shift, a.a_relu = ptr(scale), ptr(shift), int(relu)
    a.a_gate, a.gate_group = ptr(gate), gate_group
    if rowmap is not None:
        a.map_stride, a.Ho, a.Wo, a.Hi, a.Wi = rowmap
    a.W, a.ldw = ptr(W), ldw if ldw is not None else K
    a.bias, a.act = ptr(bias), act
    a.C, a.ldc, a.c_f32 = ptr(Cout), ldc if ldc is not None else N, int(Cout.dtype == torch.float32)
    if stats is not None:
        a.col_sum, a.col_sumsq = ptr(stats[0]), ptr(stats[1])
    a.M, a.N, a.K = M, N, K
    check(lib().stes_gemm(C.byref(a), stream_ptr()), "stes_gemm")


def wgrad(P, Q, dW, M, N1, N2, *, ldp=None, ldq=None, lddw=None, Q0=None, ldq0=0, k0=0,
          scale=None, shift=None, relu=False, gate=None, gate_group=0, rowmap=None, dbias=None):
    """dW += P^T pro(Q); see stes_wgrad in include/stes.h."""
    a = WgradArgs()
    a.P, a.ldp = ptr(P), ldp if ldp is not None else N1
    a.Q, a.ldq = ptr(Q), ldq if ldq is not None else N2
    a.Q0, a.ldq0, a.k0 = ptr(Q0), ldq0, k0
    a.q_scale, a.q_shift, a.q_relu = ptr(scale), ptr(shift), int(relu)
    a.q_gate, a.gate_group = ptr(gate), gate_group
    if rowmap is not None:
        a.map_stride, a.Ho, a.Wo, a.Hi, a.Wi = rowmap
    a.dW, a.lddw, a.dbias = ptr(dW), lddw if lddw is not None else N2, ptr(dbias)
    a.M, a.N1, a.N2 = M, N1, N2
    check(lib().stes_wgrad(C.byref(a), stream_ptr()), "stes_wgrad")
