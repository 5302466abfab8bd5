"""The drop-in boundary on the GPU box: libllamafile_sgemm.so called with HOST pointers exactly like
ggml calls the reference (ggml.c.patch:1957-1959, 1964, 2004-2018), checked against the oracle."""
import ctypes as C

import numpy as np
import pytest

from llamafile_amd import _hip, ggml_types as T, synth
from helpers import make_case, rel_err

pytestmark = pytest.mark.gpu


class GgmlTensor(C.Structure):
    _fields_ = [("type", C.c_int), ("backend", C.c_int), ("buffer", C.c_void_p), ("ne", C.c_int64 * 4),
                ("nb", C.c_size_t * 4), ("op", C.c_int), ("op_params", C.c_int32 * 16), ("flags", C.c_int32),
                ("grad", C.c_void_p), ("src", C.c_void_p * 10), ("view_src", C.c_void_p), ("view_offs", C.c_size_t),
                ("data", C.c_void_p), ("name", C.c_char * 128), ("extra", C.c_void_p)]


class ComputeParams(C.Structure):
    _fields_ = [("ith", C.c_int), ("nth", C.c_int), ("wsize", C.c_size_t), ("wdata", C.c_void_p), ("shared", C.c_void_p)]


@pytest.fixture(scope="module")
def host(gpu):
    lib = C.CDLL(_hip.HOST_SO)
    lib.llamafile_sgemm.restype = C.c_bool
    lib.llamafile_sgemm.argtypes = [C.c_long] * 3 + [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_long] + \
        [C.c_int] * 5
    lib.llamafile_mixmul.restype = C.c_bool
    lib.llamafile_mixmul.argtypes = [C.c_void_p] * 5
    lib.llamafile_mixmul_iqk.restype = C.c_bool
    lib.llamafile_mixmul_iqk.argtypes = [C.c_long, C.c_long, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_long, C.c_long, C.c_void_p, C.c_int, C.c_int]
    lib.llamafile_sgemm_amd_available.restype = C.c_int
    lib.llamafile_sgemm_amd_error.restype = C.c_char_p
    assert lib.llamafile_sgemm_amd_available() == 1, lib.llamafile_sgemm_amd_error()
    return lib


def tensor(arr, t, ne, nb):
    g = GgmlTensor()
    g.type = t
    for i in range(4):
        g.ne[i] = ne[i] if i < len(ne) else 1
        g.nb[i] = nb[i] if i < len(nb) else nb[-1] * (ne[len(nb) - 1] if i == len(nb) else 1)
    g.data = arr.ctypes.data
    return g


@pytest.mark.parametrize("t", [T.Q4_K, T.Q6_K, T.Q8_0, T.Q4_0, T.Q5_K], ids=lambda t: T.NAMES[t])
@pytest.mark.parametrize("n", [1, 5, 40])
def test_llamafile_sgemm_host_pointers(host, oracle, t, n):
    m, k = 96, 1024
    A, B, bt = make_case(t, m, n, k, seed=900 + t)
    ldc = m + 8  # leading-dimension slack, NaN prefill like sgemm_matmul_test.cpp:53-56
    Cm = np.full((n, ldc), np.nan, dtype=np.float32)
    kb = k // T.BLCK[t]
    nth = 3
    rets = [host.llamafile_sgemm(m, n, kb, A.ctypes.data, kb, B.ctypes.data, kb, Cm.ctypes.data, ldc, ith, nth, t, bt, T.F32)
            for ith in range(nth)]
    assert rets == [True] * nth  # every thread gets the same answer
    assert np.isnan(Cm[:, m:]).all(), "bytes outside the m x n result were touched"
    v = oracle.variant("zen4" if gpu_flags() & _hip.FLAG_Q0_VREGS32 else "avx2")
    ok, G = oracle.sgemm(t, A, bt, B, m, n, k, v=v)
    assert ok == 1
    got = Cm[:, :m]
    if t == T.Q8_0:
        assert np.array_equal(got.view(np.uint32), G.view(np.uint32))
    else:
        assert rel_err(got, G) <= (1e-3 if (t in (T.Q4_K, T.Q5_K, T.Q6_K) and n > 8) else 2e-6)  # batches: scaled f16 operands
    # second call hits the device weight cache
    Cm2 = np.full((n, ldc), np.nan, dtype=np.float32)
    assert host.llamafile_sgemm(m, n, kb, A.ctypes.data, kb, B.ctypes.data, kb, Cm2.ctypes.data, ldc, 0, 1, t, bt, T.F32)
    assert np.array_equal(Cm2[:, :m], got)


def gpu_flags():
    from llamafile_amd import sgemm
    return sgemm.host_variant_flags()


def test_llamafile_sgemm_declines_like_the_reference(host):
    A = np.zeros((4, 144), dtype=np.uint8)
    B = np.zeros((1, 4096), dtype=np.uint8)
    Cm = np.zeros((1, 4), dtype=np.float32)
    # quantised weights with f32 activations: WANT_QUANTIZATION -> false (tinyblas_cpu_sgemm.inc:183-187)
    assert not host.llamafile_sgemm(4, 1, 1, A.ctypes.data, 1, B.ctypes.data, 256, Cm.ctypes.data, 4, 0, 1, T.Q4_K, T.F32, T.F32)
    # only F32 outputs (:324-330)
    assert not host.llamafile_sgemm(4, 1, 1, A.ctypes.data, 1, B.ctypes.data, 1, Cm.ctypes.data, 4, 0, 1, T.Q4_K, T.Q8_K, T.F16)
    # unknown weight type
    assert not host.llamafile_sgemm(4, 1, 1, A.ctypes.data, 1, B.ctypes.data, 1, Cm.ctypes.data, 4, 0, 1, 16, T.Q8_K, T.F32)


@pytest.mark.parametrize("wt", [T.Q4_0, T.Q8_0, T.Q4_K], ids=lambda t: T.NAMES[t])
@pytest.mark.parametrize("tokens,tasks", [(1, 1), (5, 2), (17, 1)])
def test_llamafile_mixmul(host, oracle, wt, tokens, tasks):
    """MoE through the reference's tensor ABI: weights[cols,rows,experts], thought f32
    [cols,tasks,tokens], plan i32 [thinkers,tokens], result f32 [rows,thinkers,tokens]."""
    cols, rows, experts, thinkers = 512, 64, 8, 2
    rng = np.random.default_rng(5)
    W = np.stack([synth.random_weights(wt, rows, cols, 40 + e) for e in range(experts)])  # [experts, rows, row_bytes]
    thought = synth.random_activations(tokens * tasks, cols, 6).reshape(tokens, tasks, cols)
    plan = np.stack([rng.permutation(experts)[:thinkers] for _ in range(tokens)]).astype(np.int32)
    res = np.full((tokens, thinkers, rows), np.nan, dtype=np.float32)
    rb = W.shape[2]
    wt_t = tensor(W, wt, (cols, rows, experts), (T.TYPE_SIZE[wt], rb, rb * rows))
    th_t = tensor(thought, T.F32, (cols, tasks, tokens), (4, cols * 4, cols * 4 * tasks))
    pl_t = tensor(plan, T.I32, (thinkers, tokens), (4, 4 * thinkers))
    rs_t = tensor(res, T.F32, (rows, thinkers, tokens), (4, rows * 4, rows * 4 * thinkers))
    rets = []
    for ith in range(2):
        p = ComputeParams(ith, 2, 0, None, None)
        rets.append(host.llamafile_mixmul(C.byref(p), C.byref(wt_t), C.byref(th_t), C.byref(pl_t), C.byref(rs_t)))
    assert rets == [True, True]
    assert not np.isnan(res).any()
    if wt in (T.Q4_0, T.Q8_0):
        v = oracle.variant("zen4" if gpu_flags() & _hip.FLAG_Q0_VREGS32 else "avx2")
        ok, G = oracle.mixmul(wt, W, cols, rows, experts, thought, plan, v=v)
        assert ok == 1
    else:
        # the reference's llamafile_mixmul declines Q4_K; ggml then quantises to Q8_K and calls
        # llamafile_mixmul_iqk per expert — restate that with the oracle's iqk path
        G = np.zeros_like(res)
        q = oracle.quantize(T.Q8_K, thought.reshape(-1, cols))
        for tk in range(tokens):
            for th in range(thinkers):
                e = plan[tk, th]
                ok, c = oracle.sgemm(wt, W[e], T.Q8_K, q[tk * tasks + th % tasks][None, :], rows, 1, cols)
                assert ok == 1
                G[tk, th] = c[0]
    if wt == T.Q8_0:
        assert np.array_equal(res.view(np.uint32), G.view(np.uint32))
    else:
        # Q4_K experts in batches (> 4 tokens) run the grouped MFMA launch on scaled operands (1e-3, include/lfamd_hip.h)
        assert rel_err(res, G) <= (1e-3 if wt == T.Q4_K and tokens > 4 else 2e-6)


def test_llamafile_mixmul_iqk_row_mapping(host, oracle):
    """Per-expert call with mmid_row_mapping (ggml.c.patch:2004-2018; iqk_mul_mat.inc:84-101)."""
    t, ne00, Nx, ne11, tokens, thinkers = T.Q4_K, 512, 64, 2, 6, 2
    A = synth.random_weights(t, Nx, ne00, 77)
    x = synth.random_activations(ne11 * tokens, ne00, 78)
    B = synth.quantize_activations(T.Q8_K, x)  # wdata: rows (i11 + i12*ne11)
    mapping = np.array([[0, 1], [1, 3], [0, 4], [1, 5]], dtype=np.int32)  # {expert slot i1, token i2}
    Ny = mapping.shape[0]
    nb1, nb2 = Nx * 4, Nx * 4 * thinkers
    Cm = np.full((tokens, thinkers, Nx), np.nan, dtype=np.float32)
    G = Cm.copy()
    assert host.llamafile_mixmul_iqk(Nx, Ny, ne00, ne11, t, A.ctypes.data, B.ctypes.data, Cm.ctypes.data, nb1, nb2,
                                     mapping.ctypes.data, 0, 1)
    assert oracle.iqk_moe(t, A, B, G, Nx, Ny, ne00, ne11, nb1, nb2, mapping) == 1
    assert np.array_equal(np.isnan(Cm), np.isnan(G))  # only mapped rows are written
    mask = ~np.isnan(G)
    assert rel_err(Cm[mask], G[mask]) <= 2e-6
