"""CPU checks of the oracle itself (it is only trusted as far as these go — see DESIGN.md §2)."""
import numpy as np
import pytest

from llamafile_amd import ggml_types as T, synth
from helpers import make_case, rel_err


def test_f32_path_pinned_like_the_reference_harness(oracle):
    """The reference's own criterion for this path (sgemm_matmul_test.cpp:43-95, thresholds in its comments):
    llamafile_sgemm F32 vs a double-accumulator GEMM (ansiBLAS), NaN prefill, leading-dimension slack, ULP
    statistics.  Reduced k; the ruler-stack accumulation keeps the average error at a few ULP."""
    m, n, k = 64, 50, 26000 + 5
    lda = ldb = (k + 15) // 16 * 16
    rng = np.random.default_rng(1)
    A = np.full((m, lda), np.nan, dtype=np.float32)
    B = np.full((n, ldb), np.nan, dtype=np.float32)
    A[:, :k] = rng.random((m, k), dtype=np.float32) * 2 - 1
    B[:, :k] = rng.random((n, k), dtype=np.float32) * 2 - 1
    import ctypes as C
    L = oracle.lib()
    ldc = (m + 15) // 16 * 16
    G = np.full((n, ldc), np.nan, dtype=np.float32)
    L.ora_ansiblas_sgemm(m, n, k, A.ctypes.data, lda, B.ctypes.data, ldb, G.ctypes.data, ldc)
    for variant in ("zen4", "avx2"):
        v = oracle.variant(variant)
        Cm = np.full((n, ldc), np.nan, dtype=np.float32)
        for ith in range(3):
            r = L.ora_llamafile_sgemm(m, n, k, A.ctypes.data, lda, B.ctypes.data, ldb, Cm.ctypes.data, ldc, ith, 3, T.F32,
                                      T.F32, T.F32, C.byref(v))
            assert r == 1
        assert not np.isnan(Cm[:, :m]).any() and np.isnan(Cm[:, m:]).all()
        ulp = np.abs(Cm[:, :m].view(np.int32).astype(np.int64) - G[:, :m].view(np.int32).astype(np.int64))
        flips = (np.signbit(Cm[:, :m]) != np.signbit(G[:, :m])).sum()
        assert ulp.mean() < 100 and flips <= 0.01 * m * n  # the reference prints 94.8 ULP avg for one accumulator


@pytest.mark.parametrize("t", T.QUANT_WEIGHT_TYPES, ids=lambda t: T.NAMES[t])
def test_integer_dot_form_agrees_with_dequantised_f64(oracle, t):
    """Two independently derived restatements: integer block dots + f32 scales (from iqk_mul_mat.inc /
    tinyblas_cpu.h) vs dequantise-to-f64 (from the GPU dequantize kernels, ggml-cuda.cu.patch:3217-3471)."""
    m, n, k = 48, 4, 768
    A, B, bt = make_case(t, m, n, k, seed=40 + t)
    ok, Cm = oracle.sgemm(t, A, bt, B, m, n, k, nth=2)
    assert ok == 1 and not np.isnan(Cm).any()
    G = oracle.f64_gemm(t, A, bt, B, m, n, k)
    assert rel_err(Cm, G) <= (2e-5 if t in (T.Q4_1, T.Q5_1) else 5e-7)


@pytest.mark.parametrize("vdt", [T.Q8_0, T.Q8_1, T.Q8_K], ids=lambda t: T.NAMES[t])
def test_numpy_quantisers_equal_oracle(oracle, vdt):
    x = synth.random_activations(5, 1024, 3)
    x[1, :256] = 0
    assert np.array_equal(synth.quantize_activations(vdt, x), oracle.quantize(vdt, x))


def test_thread_partition_is_a_partition(oracle):
    """Every output is written by exactly one (ith) of nth, for both kernels' partition rules."""
    for t, (m, n, k) in ((T.Q4_K, (70, 11, 512)), (T.Q8_0, (37, 5, 128))):
        A, B, bt = make_case(t, m, n, k, seed=9)
        ok1, C1 = oracle.sgemm(t, A, bt, B, m, n, k, nth=1)
        ok3, C3 = oracle.sgemm(t, A, bt, B, m, n, k, nth=3)
        assert ok1 == ok3 == 1
        assert np.array_equal(C1.view(np.uint32), C3.view(np.uint32))


def test_q0_precise_map_geometry(oracle):
    """mnpack of tinyBLAS_Q0_AVX2 (tinyblas_cpu.h:794-931): 32-vreg builds use Kahan only on edge tiles,
    16-vreg builds never (unless --precise)."""
    z = oracle.variant("zen4")
    mode = oracle.q0_precise_map(10, 7, z)
    assert (mode[:6, :9] == 0).all()          # 3x3 region: plain
    assert mode[6, 0] == 1 and mode[0, 9] == 1  # leftovers: 1-wide tiles are Kahan
    assert (oracle.q0_precise_map(10, 7, oracle.variant("avx2")) == 0).all()
    assert (oracle.q0_precise_map(9, 1, z) == 1).all()  # decode (n = 1) on AVX512 hosts: all Kahan
    p = oracle.q0_precise_map(6, 3, oracle.variant("avx2", precise=1))
    assert (p[0] == 1).all() and (p[1:] == 0xFF).all()  # reference quirk: columns 1.. never written


def test_sgemm_declines_like_the_reference(oracle):
    A = np.zeros((4, 144), dtype=np.uint8)
    B32 = np.zeros((1, 1024), dtype=np.uint8)
    assert oracle.sgemm(T.Q4_K, A, T.F32, B32, 4, 1, 256)[0] == 0  # WANT_QUANTIZATION
    assert oracle.sgemm(T.Q8_K, np.zeros((4, 292), dtype=np.uint8), T.Q8_K, np.zeros((1, 292), dtype=np.uint8), 4, 1, 256)[0] == 0


def test_mixmul_matches_per_token_sgemm(oracle):
    cols, rows, experts, thinkers, tokens, tasks = 256, 24, 4, 2, 5, 2
    rng = np.random.default_rng(2)
    W = np.stack([synth.random_weights(T.Q8_0, rows, cols, 10 + e) for e in range(experts)])
    thought = synth.random_activations(tokens * tasks, cols, 4).reshape(tokens, tasks, cols)
    plan = np.stack([rng.permutation(experts)[:thinkers] for _ in range(tokens)]).astype(np.int32)
    ok, res = oracle.mixmul(T.Q8_0, W, cols, rows, experts, thought, plan, v=oracle.variant("avx2"))
    assert ok == 1 and not np.isnan(res).any()
    q = oracle.quantize(T.Q8_0, thought.reshape(-1, cols))
    for tk in range(tokens):
        for th in range(thinkers):
            _, c = oracle.sgemm(T.Q8_0, W[plan[tk, th]], T.Q8_0, q[tk * tasks + th % tasks][None, :], rows, 1, cols,
                                v=oracle.variant("avx2"))
            assert np.array_equal(c[0], res[tk, th])
    assert oracle.mixmul(T.Q4_K, np.zeros((2, 4, 144), dtype=np.uint8), 256, 4, 2, thought[:, :, :256], plan % 2)[0] == 0
