"""ctypes binding of libllamafile_amd_hip.so (include/lfamd_hip.h).

There is no fallback: if the shared library is missing or a call fails, an
exception is raised.  Build with ``python -c 'import __graft_entry__ as g; g.build()'``
or ``make -C llamafile_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_SO = os.environ.get("LFAMD_HIP_SO") or os.path.join(_HERE, "libllamafile_amd_hip.so")  # override: development builds
HOST_SO = os.path.join(_HERE, "libllamafile_sgemm.so")

FLAG_Q0_VREGS32 = 1
FLAG_PRECISE = 2
FLAG_FORCE_GENERIC = 4
FLAG_GEMM_NARROW = 8
FLAG_GEMM_WIDE = 16
FLAG_GEMM_PLAIN = 32
FLAG_Q80_EXACT = 64
TYPE_STAGED_Q8K = 0x1000  # the int8 batch body's staged activation image (lfamd_hip.h)
TYPE_STAGED_SCALED = 0x1001  # the scaled-operand f16 batch bodies' staged activation image


class LfamdError(RuntimeError):
    pass


_lib = None

_vp, _sz, _l, _i, _u = C.c_void_p, C.c_size_t, C.c_long, C.c_int, C.c_uint

_SIGS = {
    "lfamd_abi_version": (_i, []),
    "lfamd_last_error": (C.c_char_p, []),
    "lfamd_device_count": (_i, []),
    "lfamd_init": (_i, [_i]),
    "lfamd_device_name": (_i, [_i, C.c_char_p, _sz]),
    "lfamd_malloc": (_i, [C.POINTER(_vp), _sz]),
    "lfamd_free": (_i, [_vp]),
    "lfamd_memcpy_h2d": (_i, [_vp, _vp, _sz, _vp]),
    "lfamd_memcpy_d2h": (_i, [_vp, _vp, _sz, _vp]),
    "lfamd_memset": (_i, [_vp, _i, _sz, _vp]),
    "lfamd_stream_sync": (_i, [_vp]),
    "lfamd_packed_size": (_sz, [_i, _l, _l]),
    "lfamd_pack_weights": (_i, [_i, _l, _l, _vp, _sz, _vp, _vp]),
    "lfamd_scaled_gemm_ok": (_i, [_i, _l, _l, _vp, _vp]),
    "lfamd_mul_mat_is_exact": (_i, [_i, _l, _l, _l, _u]),
    "lfamd_mul_mat_takes_staged": (_i, [_i, _l, _l, _l, _u]),
    "lfamd_staged_q8k_size": (_sz, [_l, _l]),
    "lfamd_mul_mat_takes_staged_scaled": (_i, [_i, _l, _l, _l, _u]),
    "lfamd_staged_scaled_size": (_sz, [_l, _l]),
    "lfamd_quantize_rows": (_i, [_i, _vp, _l, _l, _sz, _vp, _sz, _vp]),
    "lfamd_mul_mat_workspace": (_sz, [_i, _l, _l, _l]),
    "lfamd_mul_mat": (_i, [_i, _vp, _l, _l, _i, _vp, _sz, _l, _vp, _l, _vp, _sz, _u, _vp]),
    "lfamd_mul_mat_multi": (_i, [_i, _i, _vp, _vp, _l, _i, _vp, _sz, _l, _vp, _vp, _vp, _sz, _u, _vp]),
    "lfamd_mul_mat_multi_types": (_i, [_i, _vp, _vp, _vp, _l, _i, _vp, _sz, _l, _vp, _vp, _vp, _sz, _u, _vp]),
    "lfamd_mul_mat_id_workspace": (_sz, [_i, _l, _l, _i, _l, _i]),
    "lfamd_mul_mat_id": (_i, [_i, _vp, _l, _l, _i, _i, _vp, _sz, _i, _l, _vp, _i, _vp, _vp, _sz, _u, _vp]),
    "lfamd_vendor_gemm_available": (_i, []),
    "lfamd_mul_mat_id_multi": (_i, [_i, _i, _vp, _l, _l, _i, _i, _vp, _sz, _i, _l, _vp, _i, _vp, _vp, _sz, _u, _vp]),
    "lfamd_rms_norm_quantize": (_i, [_vp, _sz, _vp, C.c_float, _l, _l, _i, _vp, _sz, _vp, _sz, _vp]),
    "lfamd_swiglu_quantize": (_i, [_vp, _sz, _vp, _sz, _l, _l, _i, _vp, _sz, _vp, _sz, _vp]),
    "lfamd_gemm_strided_batched_f16": (_i, [_l, _l, _l, C.c_float, _vp, _l, C.c_longlong, _vp, _l, C.c_longlong, C.c_float, _vp, _i, _l,
                                            C.c_longlong, _i, _vp]),
    "lfamd_gemm_batched_f16": (_i, [_l, _l, _l, C.c_float, _vp, _l, _vp, _l, C.c_float, _vp, _i, _l, _i, _vp]),
    "lfamd_comm_unique_id": (_i, [_vp]),
    "lfamd_comm_init": (_i, [C.POINTER(_vp), _i, _i, _vp]),
    "lfamd_comm_destroy": (_i, [_vp]),
    "lfamd_comm_init_all": (_i, [C.POINTER(_vp), _i, C.POINTER(_i), _sz]),
    "lfamd_oneshot_bytes": (_sz, [_sz]),
    "lfamd_oneshot_alloc": (_i, [C.POINTER(_vp), _sz]),
    "lfamd_oneshot_free": (_i, [_vp]),
    "lfamd_oneshot_export": (_i, [_vp, _vp]),
    "lfamd_oneshot_attach": (_i, [_vp, _vp, _sz, _vp, _sz]),
    "lfamd_comm_allreduce_add_f32": (_i, [_vp, _vp, _vp, _vp, _l, _vp]),
    "lfamd_comm_allreduce_sum_f32": (_i, [_vp, _vp, _l, _vp]),
    "lfamd_comm_allgather": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "lfamd_comm_check": (_i, [_vp]),
    "lfamd_comm_clear_error": (_i, [_vp]),
    "lfamd_time_mul_mat": (_i, [_i, _vp, _l, _l, _i, _vp, _sz, _l, _vp, _l, _vp, _sz, _u, _vp, _i, _i,
                                C.POINTER(C.c_float)]),
}

EXPORTS = tuple(_SIGS)


def lib():
    """Load the HIP module; raises LfamdError (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        # PyTorch bundles its own libamdhip64/libhsa-runtime64; a process must not end up with two HSA
        # runtimes, so when torch is the device-memory provider make sure ITS runtime is the one already
        # loaded before our module's DT_NEEDED libamdhip64.so.7 is resolved.
        if os.environ.get("LFAMD_NO_TORCH") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        if not os.path.exists(HIP_SO):
            raise LfamdError(f"{HIP_SO} not built: the MI355X HIP module is required (no CPU fallback)")
        L = C.CDLL(HIP_SO)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().lfamd_last_error().decode(errors="replace")
        raise LfamdError(f"{what} failed (status {rc}): {msg}")
