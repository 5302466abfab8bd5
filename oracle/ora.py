"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

May be imported only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg (as the checker / reported CPU baseline).  The product package
``llamafile_amd`` never imports this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


class Variant(C.Structure):
    _fields_ = [("vector_registers", C.c_int), ("kn", C.c_int), ("precise", C.c_int),
                ("kahan_contract", C.c_int)]


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("oracle.c", "oracle_bench.c", "oracle.h", "Makefile")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        vp = C.POINTER(Variant)
        L.ora_variant_zen4.restype = Variant
        L.ora_variant_avx2.restype = Variant
        L.ora_fp16_to_fp32.restype = C.c_float
        L.ora_fp16_to_fp32.argtypes = [C.c_uint16]
        L.ora_fp32_to_fp16.restype = C.c_uint16
        L.ora_fp32_to_fp16.argtypes = [C.c_float]
        for q in ("ora_quantize_row_q8_0", "ora_quantize_row_q8_1", "ora_quantize_row_q8_K"):
            getattr(L, q).argtypes = [C.c_void_p, C.c_void_p, C.c_long]
            getattr(L, q).restype = None
        L.ora_dequantize_row.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_long]
        L.ora_llamafile_sgemm.argtypes = [C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_long, C.c_void_p,
                                          C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, vp]
        L.ora_sgemm_openmp.argtypes = [C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_long, C.c_void_p,
                                       C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int,
                                       C.c_int, vp]
        L.ora_iqk_mul_mat.argtypes = [C.c_long, C.c_long, C.c_long, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_long, C.c_int, C.c_int]
        L.ora_iqk_mul_mat_moe.argtypes = [C.c_long, C.c_long, C.c_long, C.c_int, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_void_p,
                                          C.c_int, C.c_int]
        L.ora_q0_gemm.argtypes = [C.c_long, C.c_long, C.c_long, C.c_int, C.c_void_p, C.c_long, C.c_void_p,
                                  C.c_long, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_int,
                                  C.c_int, vp]
        L.ora_q0_precise_map.argtypes = [C.c_long, C.c_long, vp, C.c_void_p]
        L.ora_q0_precise_map.restype = None
        L.ora_float_gemm.argtypes = [C.c_long, C.c_long, C.c_long, C.c_int, C.c_void_p, C.c_long, C.c_int,
                                     C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int, vp]
        L.ora_ansiblas_sgemm.argtypes = [C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_long, C.c_void_p,
                                         C.c_long, C.c_void_p, C.c_long]
        L.ora_ansiblas_sgemm.restype = None
        L.ora_f64_gemm.argtypes = [C.c_long, C.c_long, C.c_long, C.c_int, C.c_void_p, C.c_size_t, C.c_int,
                                   C.c_void_p, C.c_size_t, C.c_void_p, C.c_long]
        L.ora_mixmul.argtypes = [C.c_int, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_size_t, C.c_size_t,
                                 C.c_void_p, C.c_int, C.c_long, C.c_void_p, C.c_int, C.c_void_p, vp]
        L.ora_ulp_diff.argtypes = [C.c_float, C.c_float]
        L.ora_ulp_diff.restype = C.c_longlong
        L.ora_max_threads.restype = C.c_int
        _lib = L
    return _lib


def variant(name: str = "zen4", precise: int = 0, kahan_contract: int = 1) -> Variant:
    v = lib().ora_variant_zen4() if name == "zen4" else lib().ora_variant_avx2()
    v.precise = precise
    v.kahan_contract = kahan_contract
    return v


def _p(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def quantize(vec_dot_type: int, x: np.ndarray) -> np.ndarray:
    from llamafile_amd import ggml_types as T
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, k = x.shape
    out = np.zeros((n, T.row_size(vec_dot_type, k)), dtype=np.uint8)
    fn = {T.Q8_0: lib().ora_quantize_row_q8_0, T.Q8_1: lib().ora_quantize_row_q8_1,
          T.Q8_K: lib().ora_quantize_row_q8_K}[vec_dot_type]
    for i in range(n):
        fn(x[i].ctypes.data_as(C.c_void_p), out[i].ctypes.data_as(C.c_void_p), k)
    return out


def dequantize(t: int, raw: np.ndarray, k: int) -> np.ndarray:
    rows = raw.shape[0]
    out = np.zeros((rows, k), dtype=np.float32)
    for i in range(rows):
        ok = lib().ora_dequantize_row(t, raw[i].ctypes.data_as(C.c_void_p), out[i].ctypes.data_as(C.c_void_p), k)
        assert ok
    return out


def sgemm(Atype: int, A: np.ndarray, Btype: int, B: np.ndarray, m: int, n: int, kelems: int,
          v: Variant | None = None, nth: int = 1, ldc: int | None = None):
    """llamafile_sgemm over all ith in [0,nth); A/B are [rows, row_bytes] uint8.  Returns
    (serviced, C[n, ldc] float32 NaN-prefilled)."""
    from llamafile_amd import ggml_types as T
    v = v or variant()
    ldc = ldc or m
    Cm = np.full((n, ldc), np.nan, dtype=np.float32)
    kb = kelems // T.BLCK[Atype]
    lda = A.shape[1] // T.TYPE_SIZE[Atype]
    ldb = B.shape[1] // T.TYPE_SIZE[Btype]
    ok = 1
    for ith in range(nth):
        r = lib().ora_llamafile_sgemm(m, n, kb, _p(A), lda, _p(B), ldb, _p(Cm), ldc, ith, nth, Atype, Btype,
                                      T.F32, C.byref(v))
        ok = min(ok, r)
    return ok, Cm


def sgemm_openmp(Atype, A, Btype, B, m, n, kelems, nth, v=None):
    from llamafile_amd import ggml_types as T
    v = v or variant()
    Cm = np.full((n, m), np.nan, dtype=np.float32)
    kb = kelems // T.BLCK[Atype]
    lda = A.shape[1] // T.TYPE_SIZE[Atype]
    ldb = B.shape[1] // T.TYPE_SIZE[Btype]
    r = lib().ora_sgemm_openmp(m, n, kb, _p(A), lda, _p(B), ldb, _p(Cm), m, nth, Atype, Btype, T.F32, C.byref(v))
    return r, Cm


def f64_gemm(Atype, A, Btype, B, m, n, kelems) -> np.ndarray:
    Cm = np.zeros((n, m), dtype=np.float64)
    ok = lib().ora_f64_gemm(m, n, kelems, Atype, _p(A), A.shape[1], Btype, _p(B), B.shape[1], _p(Cm), m)
    assert ok
    return Cm


def q0_precise_map(m: int, n: int, v: Variant) -> np.ndarray:
    mode = np.zeros((n, m), dtype=np.uint8)
    lib().ora_q0_precise_map(m, n, C.byref(v), _p(mode))
    return mode


def iqk_moe(typeA, A, B, Cbuf, Nx, Ny, ne00, ne11, nb1, nb2, mapping, nth=1):
    ok = 1
    for ith in range(nth):
        ok = min(ok, lib().ora_iqk_mul_mat_moe(Nx, Ny, ne00, ne11, typeA, _p(A), _p(B), _p(Cbuf), nb1, nb2,
                                               _p(mapping), ith, nth))
    return ok


def mixmul(wtype, W, cols, rows, experts, thought, plan, v=None):
    """W: uint8 [experts, rows, row_bytes]; thought f32 [tokens, tasks, cols]; plan i32 [tokens, thinkers]."""
    v = v or variant()
    tokens, tasks, _ = thought.shape
    thinkers = plan.shape[1]
    res = np.full((tokens, thinkers, rows), np.nan, dtype=np.float32)
    r = lib().ora_mixmul(wtype, _p(W), cols, rows, experts, W.shape[2], W.shape[1] * W.shape[2],
                         _p(np.ascontiguousarray(thought, dtype=np.float32)), tasks, tokens,
                         _p(np.ascontiguousarray(plan, dtype=np.int32)), thinkers, _p(res), C.byref(v))
    return r, res
