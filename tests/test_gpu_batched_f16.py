"""F16 strided-batched / pointer-array GEMM (attention KQ, KQV) against a double-accumulator reference — the criterion of the
reference's own harness for this routine (llamafile/tinyblas_test.cu:192-228 test_gsbe: gsberef<double>, sad <= 1e-4 for
f32 results; half results: 1e-2 as in checkTinyblasWorksHHHH, :21-58)."""
import ctypes as C

import numpy as np
import pytest
import torch

from llamafile_amd import _hip, ggml_types as T

pytestmark = pytest.mark.gpu


def sad(want, got):
    same = np.signbit(want) == np.signbit(got)
    return float(np.abs(want[same] - got[same]).mean()), int((~same).sum())


CASES = [
    # (m, n, k, batch, lda_pad, ldb_pad, ldc_pad): KQ (m = n_kv, k = head_dim), KQV (m = head_dim, k = n_kv), edges
    (300, 17, 128, 8, 0, 0, 0), (128, 17, 304, 8, 0, 0, 0), (128, 1, 300, 4, 0, 0, 0), (64, 64, 64, 3, 8, 8, 0),
    (33, 65, 23, 2, 1, 3, 5), (1, 1, 1, 1, 0, 0, 0), (257, 129, 1024, 2, 0, 0, 1), (2048, 512, 128, 32, 0, 0, 0),
]


@pytest.mark.parametrize("ctype", [T.F32, T.F16], ids=["f32out", "f16out"])
@pytest.mark.parametrize("case", CASES, ids=str)
@pytest.mark.parametrize("ab", [(1.0, 0.0), (0.5, 0.1)], ids=["a1b0", "a.5b.1"])
def test_strided_batched(gpu, case, ctype, ab):
    m, n, k, batch, pa, pb, pc = case
    alpha, beta = ab
    lda, ldb, ldc = k + pa, k + pb, m + pc
    rng = np.random.default_rng(m * 7 + n * 3 + k)
    A = rng.uniform(-1, 1, (batch, m, lda)).astype(np.float16)
    B = rng.uniform(-1, 1, (batch, n, ldb)).astype(np.float16)
    C0 = rng.uniform(-1, 1, (batch, n, ldc)).astype(np.float16 if ctype == T.F16 else np.float32)
    want = alpha * np.einsum("bik,bjk->bji", A[:, :, :k].astype(np.float64), B[:, :, :k].astype(np.float64))
    if beta:
        want = want + beta * C0[:, :, :m].astype(np.float64)
    dA, dB, dC = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda(), torch.from_numpy(C0.copy()).cuda()
    L = _hip.lib()
    rc = L.lfamd_gemm_strided_batched_f16(m, n, k, alpha, C.c_void_p(dA.data_ptr()), lda, m * lda, C.c_void_p(dB.data_ptr()), ldb,
                                          n * ldb, beta, C.c_void_p(dC.data_ptr()), ctype, ldc, n * ldc, batch,
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _hip.check(rc, "lfamd_gemm_strided_batched_f16")
    got = dC.cpu().numpy().astype(np.float64)
    if pc:
        assert np.array_equal(got[:, :, m:], C0[:, :, m:].astype(np.float64)), "wrote outside the m x n result"
    s, flips = sad(want, got[:, :, :m])
    assert not np.isnan(got).any()
    assert s <= (1e-4 if ctype == T.F32 else 1e-2), s
    assert flips < max(1, m * n * batch * 0.01)


def test_pointer_array_with_broadcast(gpu):
    """tinyblasGemmBatchedEx as ggml uses it for grouped-query attention: several query heads share one K head
    (k_compute_batched_ptrs, ggml-cuda.cu.patch:18200-18230: src0 pointer = head / r2)."""
    m, n, k, heads, kv_heads = 96, 5, 128, 8, 2
    rng = np.random.default_rng(3)
    K = torch.from_numpy(rng.uniform(-1, 1, (kv_heads, m, k)).astype(np.float16)).cuda()
    Q = torch.from_numpy(rng.uniform(-1, 1, (heads, n, k)).astype(np.float16)).cuda()
    out = torch.zeros((heads, n, m), dtype=torch.float32, device="cuda")
    r2 = heads // kv_heads
    pa = torch.tensor([K[h // r2].data_ptr() for h in range(heads)], dtype=torch.int64, device="cuda")
    pb = torch.tensor([Q[h].data_ptr() for h in range(heads)], dtype=torch.int64, device="cuda")
    pc = torch.tensor([out[h].data_ptr() for h in range(heads)], dtype=torch.int64, device="cuda")
    rc = _hip.lib().lfamd_gemm_batched_f16(m, n, k, 1.0, C.c_void_p(pa.data_ptr()), k, C.c_void_p(pb.data_ptr()), k, 0.0,
                                           C.c_void_p(pc.data_ptr()), T.F32, m, heads, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _hip.check(rc, "lfamd_gemm_batched_f16")
    want = np.einsum("hik,hjk->hji", K.cpu().numpy().astype(np.float64)[np.arange(heads) // r2], Q.cpu().numpy().astype(np.float64))
    s, flips = sad(want, out.cpu().numpy().astype(np.float64))
    assert s <= 1e-4 and flips < heads * m * n * 0.01


def test_rejects_bad_arguments(gpu):
    L = _hip.lib()
    z = C.c_void_p(0)
    assert L.lfamd_gemm_strided_batched_f16(4, 4, 8, 1.0, z, 4, 0, z, 8, 0, 0.0, z, T.F32, 4, 0, 1, None) == -2  # lda < k
    assert L.lfamd_gemm_strided_batched_f16(4, 4, 8, 1.0, z, 8, 0, z, 8, 0, 0.0, z, T.Q4_K, 4, 0, 1, None) == -2  # result type
    assert L.lfamd_gemm_strided_batched_f16(0, 4, 8, 1.0, z, 8, 0, z, 8, 0, 0.0, z, T.F32, 4, 0, 1, None) == 0  # empty: nothing to do
