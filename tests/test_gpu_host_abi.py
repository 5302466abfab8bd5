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
    lib.llamafile_sgemm_amd_register_weights.argtypes = [C.c_void_p, C.c_size_t]
    lib.llamafile_sgemm_amd_unregister_weights.argtypes = [C.c_void_p]
    lib.llamafile_sgemm_amd_forget.argtypes = [C.c_void_p]
    lib.llamafile_sgemm_amd_cached_bytes.restype = C.c_size_t
    lib.llamafile_sgemm_amd_set_cache_budget.argtypes = [C.c_size_t]
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
    if t == T.Q8_0 and n <= 8:  # the vecdot: bit for bit; batches run the f16 MFMA body (LFAMD_Q80_EXACT=1 for the exact one)
        assert np.array_equal(got.view(np.uint32), G.view(np.uint32))
    elif t == T.Q8_0:
        from helpers import q80_batch_tol
        assert rel_err(got, G) <= q80_batch_tol(m, k, n)
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
    if wt == T.Q8_0 and tokens <= 8:  # at most 8 rows per expert: the bit-exact vecdot kernels
        assert np.array_equal(res.view(np.uint32), G.view(np.uint32))
    elif wt == T.Q8_0:  # an expert with more than 8 rows runs the MFMA body (LFAMD_Q80_EXACT=1 for the bit-exact one)
        assert rel_err(res, G) <= 2e-6
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


def test_mutable_A_is_uploaded_every_call(host, oracle):
    """ggml also enters llamafile_sgemm with the KV cache as A (KQ, KQV): same address, same shape, bytes changed in
    the MIDDLE between two calls.  The device copy of unregistered, writable host memory must never be reused
    (round-1 hazard: a cache keyed by address + 3 x 64 sampled bytes)."""
    m, n, k = 96, 1, 1024
    rng = np.random.default_rng(11)
    A = rng.standard_normal((m, k)).astype(np.float16)
    B = rng.standard_normal((n, k)).astype(np.float32)
    Cm = np.zeros((n, m), dtype=np.float32)
    before = host.llamafile_sgemm_amd_cached_bytes()
    assert host.llamafile_sgemm(m, n, k, A.ctypes.data, k, B.ctypes.data, k, Cm.ctypes.data, m, 0, 1, T.F16, T.F32, T.F32)
    want = B.astype(np.float64) @ A.astype(np.float64).T
    assert rel_err(Cm, want) <= 1e-5
    assert host.llamafile_sgemm_amd_cached_bytes() == before, "writable, unregistered host memory was cached"
    A[m // 2, 300:340] = np.float16(7.0)  # an interior row: first / middle / last 64 bytes unchanged
    Cm2 = np.zeros_like(Cm)
    assert host.llamafile_sgemm(m, n, k, A.ctypes.data, k, B.ctypes.data, k, Cm2.ctypes.data, m, 0, 1, T.F16, T.F32, T.F32)
    want2 = B.astype(np.float64) @ A.astype(np.float64).T
    assert abs(want2[0, m // 2] - want[0, m // 2]) > 1.0
    assert rel_err(Cm2, want2) <= 1e-5, "stale device copy of a mutated A"
    # the same for a quantised A (quantised KV cache: -ctk q8_0)
    Aq, Bq, bt = make_case(T.Q8_0, m, 1, k, seed=5)
    Cq = np.zeros((1, m), dtype=np.float32)
    kb = k // 32
    assert host.llamafile_sgemm(m, 1, kb, Aq.ctypes.data, kb, Bq.ctypes.data, kb, Cq.ctypes.data, m, 0, 1, T.Q8_0, bt, T.F32)
    Aq[m // 2, 2 + 34 * 7:2 + 34 * 7 + 32] ^= 0x55
    v = oracle.variant("zen4" if gpu_flags() & _hip.FLAG_Q0_VREGS32 else "avx2")
    ok, G = oracle.sgemm(T.Q8_0, Aq, bt, Bq, m, 1, k, v=v)
    Cq2 = np.zeros_like(Cq)
    assert host.llamafile_sgemm(m, 1, kb, Aq.ctypes.data, kb, Bq.ctypes.data, kb, Cq2.ctypes.data, m, 0, 1, T.Q8_0, bt, T.F32)
    assert ok == 1 and np.array_equal(Cq2.view(np.uint32), G.view(np.uint32))
    assert not np.array_equal(Cq2, Cq)


def test_registered_and_readonly_weights_are_kept(host, oracle, tmp_path):
    """Device copies are kept only for immutable host bytes: a registered range, or a mapping without write permission
    (an mmap'd GGUF).  LRU eviction inside the byte budget; unregister frees."""
    t, m, n, k = T.Q4_K, 64, 1, 1024
    A, B, bt = make_case(t, m, n, k, seed=21)
    kb = k // 256
    ok, G = oracle.sgemm(t, A, bt, B, m, n, k)
    assert ok == 1

    def call(a):
        out = np.zeros((n, m), dtype=np.float32)
        assert host.llamafile_sgemm(m, n, kb, a.ctypes.data, kb, B.ctypes.data, kb, out.ctypes.data, m, 0, 1, t, bt, T.F32)
        return out

    base = host.llamafile_sgemm_amd_cached_bytes()
    assert rel_err(call(A), G) <= 2e-6
    assert host.llamafile_sgemm_amd_cached_bytes() == base  # writable numpy memory: not kept
    host.llamafile_sgemm_amd_register_weights(A.ctypes.data, A.nbytes)
    assert rel_err(call(A), G) <= 2e-6
    kept = host.llamafile_sgemm_amd_cached_bytes() - base
    assert kept == _hip.lib().lfamd_packed_size(t, m, k)
    assert rel_err(call(A), G) <= 2e-6  # served from the kept copy
    assert host.llamafile_sgemm_amd_cached_bytes() - base == kept
    host.llamafile_sgemm_amd_unregister_weights(A.ctypes.data)
    assert host.llamafile_sgemm_amd_cached_bytes() == base
    # a read-only file mapping needs no registration
    f = tmp_path / "weights.bin"
    A.tofile(f)
    Am = np.memmap(f, dtype=np.uint8, mode="r", shape=A.shape)
    assert rel_err(call(Am), G) <= 2e-6
    assert host.llamafile_sgemm_amd_cached_bytes() - base == kept
    # budget: a second tensor that does not fit evicts the least recently used one
    host.llamafile_sgemm_amd_set_cache_budget(base + kept)
    A2 = synth.random_weights(t, m, k, 22)
    host.llamafile_sgemm_amd_register_weights(A2.ctypes.data, A2.nbytes)
    ok2, G2 = oracle.sgemm(t, A2, bt, B, m, n, k)
    assert rel_err(call(A2), G2) <= 2e-6
    assert host.llamafile_sgemm_amd_cached_bytes() - base == kept
    assert rel_err(call(Am), G) <= 2e-6  # (packed again)
    host.llamafile_sgemm_amd_set_cache_budget(200 << 30)
    host.llamafile_sgemm_amd_unregister_weights(A2.ctypes.data)
    host.llamafile_sgemm_amd_forget(Am.ctypes.data)
    assert host.llamafile_sgemm_amd_cached_bytes() == base


def test_remapped_address_is_not_served_from_the_kept_copy(host, oracle, tmp_path):
    """a model is munmap()ed and ANOTHER file lands at the same address (same shape, same type).  The host never
    calls _forget / _reset; the kept copy must not be served (mapping identity + fingerprint are re-checked on a hit)."""
    import mmap
    t, m, n, k = T.Q4_K, 64, 1, 1024
    A1, B, bt = make_case(t, m, n, k, seed=31)
    A2 = synth.random_weights(t, m, k, 32)
    kb = k // 256
    G1 = oracle.sgemm(t, A1, bt, B, m, n, k)[1]
    G2 = oracle.sgemm(t, A2, bt, B, m, n, k)[1]
    assert rel_err(G1, G2) > 0.1
    f1, f2 = tmp_path / "w1.bin", tmp_path / "w2.bin"
    A1.tofile(f1)
    A2.tofile(f2)
    libc = C.CDLL(None, use_errno=True)
    libc.mmap.restype = C.c_void_p
    libc.mmap.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
    libc.munmap.argtypes = [C.c_void_p, C.c_size_t]
    size = (A1.nbytes + 4095) // 4096 * 4096

    def map_at(path, addr):
        import os
        fd = os.open(path, os.O_RDONLY)
        flags = mmap.MAP_SHARED | (0x10 if addr else 0)  # MAP_FIXED: replaces what is mapped there
        p = libc.mmap(addr, size, mmap.PROT_READ, flags, fd, 0)
        os.close(fd)
        assert p not in (None, C.c_void_p(-1).value)
        return p

    def call(addr):
        out = np.zeros((n, m), dtype=np.float32)
        assert host.llamafile_sgemm(m, n, kb, addr, kb, B.ctypes.data, kb, out.ctypes.data, m, 0, 1, t, bt, T.F32)
        return out

    base = host.llamafile_sgemm_amd_cached_bytes()
    p = map_at(str(f1), None)
    assert rel_err(call(p), G1) <= 2e-6
    kept = host.llamafile_sgemm_amd_cached_bytes() - base
    assert kept > 0
    assert rel_err(call(p), G1) <= 2e-6  # (a hit)
    assert map_at(str(f2), p) == p  # the other file, same address, no unmap seen by the library
    got = call(p)
    assert rel_err(got, G2) <= 2e-6, ("stale packed weights served for a remapped address", rel_err(got, G1),
                                      host.llamafile_sgemm_amd_cached_bytes() - base, kept)
    assert host.llamafile_sgemm_amd_cached_bytes() - base == kept
    host.llamafile_sgemm_amd_forget(p)
    libc.munmap(p, size)


def test_mutable_A_does_not_reparse_maps_every_call(host):
    """KQ / KQV enter with writable memory as A on every call: that answer comes from the snapshot (re-read at most every 200 ms)."""
    m, n, k = 32, 1, 256
    A = np.ones((m, k), dtype=np.float16)
    B = np.ones((n, k), dtype=np.float32)
    Cm = np.zeros((n, m), dtype=np.float32)
    host.llamafile_sgemm_amd_maps_reads.restype = C.c_ulong
    assert host.llamafile_sgemm(m, n, k, A.ctypes.data, k, B.ctypes.data, k, Cm.ctypes.data, m, 0, 1, T.F16, T.F32, T.F32)
    r0 = host.llamafile_sgemm_amd_maps_reads()
    import time
    t0 = time.monotonic()
    for _ in range(50):
        assert host.llamafile_sgemm(m, n, k, A.ctypes.data, k, B.ctypes.data, k, Cm.ctypes.data, m, 0, 1, T.F16, T.F32, T.F32)
    dt = time.monotonic() - t0
    assert host.llamafile_sgemm_amd_maps_reads() - r0 <= 1 + int(dt / 0.2) + 1
    assert np.allclose(Cm, k)


def test_mixmul_mutated_experts(host, oracle):
    """llamafile_mixmul with writable expert weights changed in place between two calls."""
    wt, cols, rows, experts, thinkers, tokens, tasks = T.Q8_0, 512, 64, 4, 2, 3, 1
    W = np.stack([synth.random_weights(wt, rows, cols, 60 + e) for e in range(experts)])
    thought = synth.random_activations(tokens * tasks, cols, 8).reshape(tokens, tasks, cols)
    plan = np.array([[0, 1], [2, 3], [1, 2]], dtype=np.int32)
    rb = W.shape[2]
    v = oracle.variant("zen4" if gpu_flags() & _hip.FLAG_Q0_VREGS32 else "avx2")

    def call():
        res = np.full((tokens, thinkers, rows), np.nan, dtype=np.float32)
        wt_t = tensor(W, wt, (cols, rows, experts), (T.TYPE_SIZE[wt], rb, rb * rows))
        th_t = tensor(thought, T.F32, (cols, tasks, tokens), (4, cols * 4, cols * 4 * tasks))
        pl_t = tensor(plan, T.I32, (thinkers, tokens), (4, 4 * thinkers))
        rs_t = tensor(res, T.F32, (rows, thinkers, tokens), (4, rows * 4, rows * 4 * thinkers))
        p = ComputeParams(0, 1, 0, None, None)
        assert host.llamafile_mixmul(C.byref(p), C.byref(wt_t), C.byref(th_t), C.byref(pl_t), C.byref(rs_t))
        return res

    r1 = call()
    ok, G1 = oracle.mixmul(wt, W, cols, rows, experts, thought, plan, v=v)
    assert ok == 1 and np.array_equal(r1.view(np.uint32), G1.view(np.uint32))
    W[2, rows // 2, 2 + 34 * 5:2 + 34 * 5 + 32] ^= 0x33  # expert 2, an interior row
    r2 = call()
    ok, G2 = oracle.mixmul(wt, W, cols, rows, experts, thought, plan, v=v)
    assert ok == 1 and np.array_equal(r2.view(np.uint32), G2.view(np.uint32))
    assert not np.array_equal(r1, r2)
