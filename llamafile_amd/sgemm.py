"""Host-side mirror of llamafile's matmul plug-in interface on device-resident data.

Same names, argument meaning and error behaviour as the reference
(/root/reference/llamafile/sgemm.h:23-28, sgemm.cpp:104-145):
``llamafile_sgemm`` returns ``False`` ("not serviced", the caller falls back) for
type combinations the path does not handle and raises on violated preconditions
(the reference asserts, tinyblas_cpu_sgemm.inc:277-284).  Tensors are torch CUDA
tensors used purely as device memory; every computation is a call through the C
ABI of libllamafile_amd_hip.so (include/lfamd_hip.h).  No CPU fallback exists.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _hip
from . import ggml_types as T


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


_host_flags = None


def host_variant_flags() -> int:
    """Which build of tinyBLAS_Q0 the reference would dispatch to on THIS host
    (sgemm.cpp:26-102: AVX512F -> the 32-vector-register builds), so Q8_0 results match the
    reference "on the host's own cores"."""
    global _host_flags
    if _host_flags is None:
        flags = 0
        try:
            with open("/proc/cpuinfo") as f:
                for line in f:
                    if line.startswith("flags"):
                        if " avx512f" in line:
                            flags |= _hip.FLAG_Q0_VREGS32
                        break
        except OSError:
            pass
        _host_flags = flags
    return _host_flags


@dataclass
class PackedWeights:
    """A weight tensor resident in HBM in the module's packed layout."""
    type: int
    rows: int
    cols: int
    data: torch.Tensor  # uint8, cuda
    exact_only: bool = False  # Q4_K / Q5_K block scales outside the scaled-operand GEMM's range (lfamd_scaled_gemm_ok)

    @property
    def nbytes(self) -> int:
        """ALGORITHMIC bytes: the tensor's size in the GGUF file (what one pass over the weights has to read at least).  The
        resident image may be larger (Q3_K, IQ4_XS, Q4_1 / Q5_0 / Q5_1; Q8_0 only in a process that opted into the vendor GEMM): `resident_bytes`."""
        return self.rows * T.row_size(self.type, self.cols)

    @property
    def resident_bytes(self) -> int:
        return self.data.numel()


def init(device: int = 0) -> None:
    _hip.check(_hip.lib().lfamd_init(device), "lfamd_init")
    torch.cuda.set_device(device)


def upload_weights(t: int, raw, rows: int, cols: int, device="cuda") -> PackedWeights:
    """raw: uint8 [rows, row_bytes] (numpy or torch) in GGUF layout -> packed device tensor."""
    L = _hip.lib()
    if isinstance(raw, np.ndarray):
        raw = torch.from_numpy(np.ascontiguousarray(raw))
    raw = raw.to(device, non_blocking=False).contiguous()
    assert raw.dtype == torch.uint8 and raw.dim() == 2 and raw.shape[0] == rows
    size = L.lfamd_packed_size(t, rows, cols)
    if size == 0 and rows and cols:
        raise _hip.LfamdError(f"unsupported weight type {T.NAMES.get(t, t)} or bad shape")
    out = torch.empty(max(size, 16), dtype=torch.uint8, device=raw.device)
    _hip.check(L.lfamd_pack_weights(t, rows, cols, _ptr(raw), raw.shape[1], _ptr(out), _stream()), "lfamd_pack_weights")
    W = PackedWeights(t, rows, cols, out[:size] if size else out[:0])
    if t in (T.Q4_K, T.Q5_K, T.Q6_K) and rows and cols:
        # scaled-operand batches need |d| < 64/63 (include/lfamd_hip.h): out-of-range matrices always run exact
        ok = L.lfamd_scaled_gemm_ok(t, rows, cols, _ptr(out), _stream())
        if ok < 0:
            raise _hip.LfamdError("lfamd_scaled_gemm_ok failed")
        W.exact_only = ok == 0
    return W


def quantize_rows(vec_dot_type: int, x: torch.Tensor) -> torch.Tensor:
    """f32 [n, k] cuda -> activation blocks uint8 [n, row_size]."""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
    x = x.contiguous()
    n, k = x.shape
    rb = T.row_size(vec_dot_type, k)
    y = torch.empty((n, rb), dtype=torch.uint8, device=x.device)
    _hip.check(_hip.lib().lfamd_quantize_rows(vec_dot_type, _ptr(x), n, k, k * 4, _ptr(y), rb, _stream()),
               "lfamd_quantize_rows")
    return y


def workspace_bytes(t: int, m: int, k: int, n: int) -> int:
    return int(_hip.lib().lfamd_mul_mat_workspace(t, m, k, n))


def mul_mat(W: PackedWeights, B: torch.Tensor, Btype: int, n: int | None = None, out: torch.Tensor | None = None,
            ldc: int | None = None, flags: int | None = None, workspace: torch.Tensor | None = None) -> torch.Tensor:
    """C[j, i] = sum_l W[i, l] * B[j, l]  (GGML_OP_MUL_MAT).  B: uint8 [n, b_row_bytes] in Btype blocks.
    Returns f32 [n, ldc] (column-major m x n like llamafile_sgemm's C)."""
    L = _hip.lib()
    assert B.is_cuda and B.dtype == torch.uint8 and B.dim() == 2 and B.stride(1) == 1
    n = B.shape[0] if n is None else n
    ldc = W.rows if ldc is None else ldc
    if out is None:
        out = torch.empty((n, ldc), dtype=torch.float32, device=B.device)
    flags = host_variant_flags() if flags is None else flags
    if getattr(W, "exact_only", False):
        flags |= _hip.FLAG_PRECISE
    need = L.lfamd_mul_mat_workspace(W.type, W.rows, W.cols, n)
    if need and (workspace is None or workspace.numel() < need):
        workspace = torch.empty(need, dtype=torch.uint8, device=B.device)
    ws_ptr = _ptr(workspace) if workspace is not None else C.c_void_p(0)
    ws_len = workspace.numel() if workspace is not None else 0
    rc = L.lfamd_mul_mat(W.type, _ptr(W.data), W.rows, W.cols, Btype, _ptr(B), B.stride(0), n, _ptr(out), ldc, ws_ptr,
                         ws_len, flags, _stream())
    _hip.check(rc, "lfamd_mul_mat")
    return out


def mul_mat_multi(Ws: list, B: torch.Tensor, Btype: int, n: int | None = None, flags: int | None = None,
                  workspace: torch.Tensor | None = None) -> list:
    """Several GGML_OP_MUL_MAT nodes sharing the activations B (same k): one fused launch where the module can
    (decode GEMV of one type, or of the K-quant pair {Q4_K | Q5_K, Q6_K}; batches of one K-quant type), else one per
    matrix.  Returns the list of f32 [n, m_j] outputs."""
    L = _hip.lib()
    assert len(Ws) >= 1 and all(w.cols == Ws[0].cols for w in Ws)
    mixed = any(w.type != Ws[0].type for w in Ws)
    n = B.shape[0] if n is None else n
    flags = host_variant_flags() if flags is None else flags
    if any(getattr(w, "exact_only", False) for w in Ws):
        flags |= _hip.FLAG_PRECISE
    outs = [torch.empty((n, w.rows), dtype=torch.float32, device=B.device) for w in Ws]
    need = max(L.lfamd_mul_mat_workspace(w.type, w.rows, w.cols, n) for w in Ws)
    if need and (workspace is None or workspace.numel() < need):
        workspace = torch.empty(need, dtype=torch.uint8, device=B.device)
    cnt = len(Ws)
    A_arr = (C.c_void_p * cnt)(*[w.data.data_ptr() for w in Ws])
    C_arr = (C.c_void_p * cnt)(*[o.data_ptr() for o in outs])
    m_arr = (C.c_long * cnt)(*[w.rows for w in Ws])
    ldc_arr = (C.c_long * cnt)(*[w.rows for w in Ws])
    ws_ptr = _ptr(workspace) if workspace is not None else C.c_void_p(0)
    ws_len = workspace.numel() if workspace is not None else 0
    if mixed:
        t_arr = (C.c_int * cnt)(*[w.type for w in Ws])
        rc = L.lfamd_mul_mat_multi_types(cnt, t_arr, A_arr, m_arr, Ws[0].cols, Btype, _ptr(B), B.stride(0) * B.element_size(),
                                         n, C_arr, ldc_arr, ws_ptr, ws_len, flags, _stream())
    else:
        rc = L.lfamd_mul_mat_multi(Ws[0].type, cnt, A_arr, m_arr, Ws[0].cols, Btype, _ptr(B), B.stride(0) * B.element_size(),
                                   n, C_arr, ldc_arr, ws_ptr, ws_len, flags, _stream())
    _hip.check(rc, "lfamd_mul_mat_multi")
    return outs


def llamafile_sgemm(m: int, n: int, k: int, A: PackedWeights, lda: int, B: torch.Tensor, ldb: int, Cout: torch.Tensor,
                    ldc: int, ith: int, nth: int, Atype: int, Btype: int, Ctype: int, flags: int | None = None) -> bool:
    """Device-resident mirror of llamafile_sgemm (sgemm.h:23-24).  k, lda, ldb are in BLOCKS for
    quantised types like the reference.  All ``nth`` callers get the same boolean; thread 0 does the
    launch (SURVEY.md §8 b-1 threading contract)."""
    if not (m >= 0 and n >= 0 and k >= 0 and lda >= k and ldb >= k and ldc >= m and nth > 0 and ith < nth):
        raise AssertionError("llamafile_sgemm precondition violated")  # the reference asserts
    if Ctype != T.F32 or Atype not in T.VEC_DOT:
        return False
    vdt = T.VEC_DOT[Atype]
    if Atype in (T.F32,):
        ok = Btype == T.F32
    elif Atype in (T.F16, T.BF16):
        ok = Btype in (T.F32, Atype)
    else:
        ok = Btype == vdt
    if not ok:
        return False
    if A.type != Atype or A.rows != m or A.cols != k * T.BLCK[Atype] or lda != k:
        return False  # packed weights cover whole contiguous rows only
    if ith == 0:
        mul_mat(A, B, Btype, n=n, out=Cout, ldc=ldc, flags=flags)
    return True


_moe_scaled_ok: dict = {}


def mul_mat_id(Ws: torch.Tensor, wtype: int, rows: int, cols: int, experts: int, thought: torch.Tensor, Btype: int,
               tasks: int, tokens: int, plan: torch.Tensor, thinkers: int, flags: int | None = None,
               prefill: float | None = None) -> torch.Tensor:
    """GGML_OP_MUL_MAT_ID.  Ws: packed weights of all experts back to back; thought: uint8
    [tokens*tasks, row_bytes]; plan int32 [tokens, thinkers].  Returns f32 [tokens, thinkers, rows]."""
    L = _hip.lib()
    flags = host_variant_flags() if flags is None else flags
    if wtype in (T.Q4_K, T.Q5_K, T.Q6_K) and tokens > 4:
        # batches run on scaled operands: check the stack's block scales once (cf. upload_weights)
        key = (Ws.data_ptr(), Ws.numel(), wtype)
        if key not in _moe_scaled_ok:
            ok = L.lfamd_scaled_gemm_ok(wtype, experts * ((rows + 31) // 32) * 32, cols, _ptr(Ws), _stream())
            if ok < 0:
                raise _hip.LfamdError("lfamd_scaled_gemm_ok failed")
            _moe_scaled_ok[key] = ok == 1
        if not _moe_scaled_ok[key]:
            flags |= _hip.FLAG_PRECISE
    res = torch.empty((tokens, thinkers, rows), dtype=torch.float32, device=thought.device)
    if prefill is not None:  # tests: rows of out-of-range expert ids must stay untouched
        res.fill_(prefill)
    need = L.lfamd_mul_mat_id_workspace(wtype, rows, cols, experts, tokens, thinkers)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=thought.device)
    rc = L.lfamd_mul_mat_id(wtype, _ptr(Ws), rows, cols, experts, Btype, _ptr(thought), thought.stride(0), tasks, tokens,
                            _ptr(plan), thinkers, _ptr(res), _ptr(ws), ws.numel(), flags, _stream())
    _hip.check(rc, "lfamd_mul_mat_id")
    return res


def time_mul_mat(W: PackedWeights, B: torch.Tensor, Btype: int, n: int, warmup: int = 5, iters: int = 50,
                 flags: int | None = None) -> float:
    """Average device microseconds per lfamd_mul_mat launch (HIP events on the current stream)."""
    L = _hip.lib()
    out = torch.empty((n, W.rows), dtype=torch.float32, device=B.device)
    flags = host_variant_flags() if flags is None else flags
    need = L.lfamd_mul_mat_workspace(W.type, W.rows, W.cols, n)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=B.device)
    us = C.c_float(0)
    rc = L.lfamd_time_mul_mat(W.type, _ptr(W.data), W.rows, W.cols, Btype, _ptr(B), B.stride(0), n, _ptr(out), W.rows,
                              _ptr(ws), ws.numel(), flags, _stream(), warmup, iters, C.byref(us))
    _hip.check(rc, "lfamd_time_mul_mat")
    return float(us.value)
