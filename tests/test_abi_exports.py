"""CPU-side checks of the C-ABI libraries: they load, and export every symbol the headers declare.
No compute call is made (there is no GPU here)."""
import ctypes as C
import os
import re

import numpy as np

from llamafile_amd import _hip, ggml_types as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"static inline[^{]*\{.*?\n\}", "", src, flags=re.S)
    names = re.findall(r"\b([a-z_][a-z0-9_]*)\s*\([^;{]*\)\s*;", src)
    return sorted(set(n for n in names if n.startswith(("lfamd_", "llamafile_", "iqk_"))))


def test_hip_module_exports_every_declared_symbol():
    assert os.path.exists(_hip.HIP_SO), "run __graft_entry__.build() first"
    lib = C.CDLL(_hip.HIP_SO)
    fns = declared_functions("lfamd_hip.h")
    assert len(fns) >= 18
    for name in fns:
        assert hasattr(lib, name), name
    assert set(_hip.EXPORTS) <= set(fns)
    assert lib.lfamd_abi_version() == 1


def test_host_plugin_exports_reference_abi():
    lib = C.CDLL(_hip.HOST_SO)
    fns = declared_functions("llamafile_sgemm.h")
    for name in ("llamafile_sgemm", "llamafile_mixmul", "llamafile_mixmul_needs", "llamafile_mixmul_iqk", "iqk_mul_mat",
                 "iqk_mul_mat_moe"):
        assert name in fns
    for name in fns:
        assert hasattr(lib, name), name


def test_packed_size_is_exactly_gguf_size_for_aligned_shapes():
    lib = _hip.lib()
    for t in (T.Q4_K, T.Q6_K):
        assert lib.lfamd_packed_size(t, 4096, 4096) == 4096 * T.row_size(t, 4096)
    # Q8_0: the GGUF-sized P80 image and nothing else — the vecdot GEMV, the bit-exact batch kernel and the f16 MFMA batch body all
    # read it (a process that opted into the vendor GEMM, LFAMD_USE_BLASLT=1, also keeps plain f16 rows: 2 more bytes per weight)
    lt = lib.lfamd_vendor_gemm_available()
    assert lib.lfamd_packed_size(T.Q8_0, 4096, 4096) == 4096 * T.row_size(T.Q8_0, 4096) + (4096 * 4096 * 2 if lt else 0)
    assert lib.lfamd_packed_size(T.Q8_0, 64, 96) == 64 // 8 * 1088 + (64 * 96 * 2 if lt else 0)  # 3 blocks -> one P80 tile of four
    assert lib.lfamd_packed_size(T.Q4_K, 33, 256) == 2 * 4608  # rows round up to 32
    # Q2_K / Q3_K: compact resident images (84 / 116 bytes per 256 weights; the canonical image is rebuilt per batch call);
    assert lib.lfamd_packed_size(T.Q2_K, 4096, 4096) == 128 * 16 * 2688 == 4096 * T.row_size(T.Q2_K, 4096)
    assert lib.lfamd_packed_size(T.Q3_K, 4096, 4096) == 128 * 16 * 3712
    assert lib.lfamd_packed_size(T.Q3_K, 4096, 4096) <= 1.06 * 4096 * T.row_size(T.Q3_K, 4096)
    assert lib.lfamd_packed_size(T.IQ4_XS, 4096, 4096) == 128 * 16 * 4608  # codebook indices on the nibble lattice: 144 B per 256 (file: 136)
    for t in (T.Q4_1, T.Q5_0, T.Q5_1):  # PCL when rows are whole 256-weight groups, RAW otherwise (like Q4_0 / P40)
        assert lib.lfamd_packed_size(t, 4096, 4096) == 128 * 16 * 6144
        assert lib.lfamd_packed_size(t, 64, 96) == 64 * T.row_size(t, 96)
    assert lib.lfamd_packed_size(T.Q4_K, 32, 100) == 0  # cols not a block multiple
    assert lib.lfamd_packed_size(99, 32, 256) == 0  # unknown type


def test_host_plugin_answers_false_without_a_gpu():
    """sgemm.cpp's contract: `false` = not serviced, the caller falls back.  Without an MI355X the
    plug-in must say false (never compute on the CPU, never crash)."""
    import torch
    if torch.cuda.is_available():
        return
    lib = C.CDLL(_hip.HOST_SO)
    lib.llamafile_sgemm.restype = C.c_bool
    lib.llamafile_sgemm.argtypes = [C.c_long] * 3 + [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_long] + \
        [C.c_int] * 5
    A = np.zeros((4, 144), dtype=np.uint8)
    B = np.zeros((1, 292), dtype=np.uint8)
    Cm = np.zeros((1, 4), dtype=np.float32)
    ok = lib.llamafile_sgemm(4, 1, 1, A.ctypes.data, 1, B.ctypes.data, 1, Cm.ctypes.data, 4, 0, 1, T.Q4_K, T.Q8_K, T.F32)
    assert ok is False
    lib.llamafile_sgemm_amd_available.restype = C.c_int
    assert lib.llamafile_sgemm_amd_available() == 0
    lib.llamafile_sgemm_amd_error.restype = C.c_char_p
    assert lib.llamafile_sgemm_amd_error()


def test_single_process_communicators_decline_cleanly():
    """lfamd_comm_init_all: bad arguments are refused; without a GPU it reports an error instead of crashing."""
    import torch
    lib = _hip.lib()
    comms = (C.c_void_p * 9)()
    devs = (C.c_int * 9)(*range(9))
    assert lib.lfamd_comm_init_all(comms, 0, devs, 65536) != 0
    assert lib.lfamd_comm_init_all(comms, 9, devs, 65536) != 0
    assert lib.lfamd_comm_init_all(comms, 2, devs, 0) != 0
    assert lib.lfamd_comm_init_all(None, 2, devs, 65536) != 0
    if not torch.cuda.is_available():
        assert lib.lfamd_comm_init_all(comms, 2, devs, 65536) != 0
        assert all(c is None for c in comms[:2])


def test_moe_inner_workspace_covers_every_smaller_batch():
    """The MUL_MAT_ID host path sizes ONE inner workspace for per-expert calls of any n up to tokens x thinkers (csrc/moe.hip).
    lfamd_mul_mat_workspace is NOT monotonic in n (a K-split launch of few tiles keeps partial tiles a larger batch does not need),
    so the bound used there must cover every smaller batch."""
    import ctypes as C
    lib = _hip.lib()
    upto = lib.lfamd_mul_mat_workspace_upto
    upto.restype, upto.argtypes = C.c_size_t, [C.c_int, C.c_long, C.c_long, C.c_long]
    for t in (T.Q4_K, T.Q5_K, T.Q6_K, T.Q8_0, T.Q4_0, T.Q5_1, T.Q2_K, T.Q3_K, T.IQ4_XS, T.F16, T.BF16):
        for m, k in ((4096, 4096), (14336, 4096), (4096, 14336), (1024, 512)):
            prev = 0
            for n in list(range(1, 70)) + list(range(70, 1200, 37)):
                bound = upto(t, m, k, n)
                assert bound >= prev and bound >= lib.lfamd_mul_mat_workspace(t, m, k, n), (T.NAMES[t], m, k, n)
                prev = bound
            for n in (100, 144, 300, 1100):  # every smaller batch, exhaustively, at a few sizes
                bound = upto(t, m, k, n)
                assert all(lib.lfamd_mul_mat_workspace(t, m, k, v) <= bound for v in range(1, n + 1)), (T.NAMES[t], m, k, n)
