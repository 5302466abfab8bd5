"""The reference's own GPU test harness for the float mat-mul, run against the HIP float path.

llamafile/tester.cu:66-103 (test_matmul): every (m, n, k) of kDims^3 with leading-dimension slack 0 and 1, then
128^3; llamafile/tinyblas_test.cu:141-190, 230-245: inputs numba() in (-1, 1), outputs pre-filled with a tombstone,
gemmref<double> as the expected result, and the pass rule of llamafile/tester.h:225-252 — no NaN, mean absolute
difference over the sign-agreeing outputs ("sad") <= tolerance, fewer than 1 % sign flips.  Tolerances are the
reference's: 1e-4 for f32 in / f32 out (checkTinyblasWorksSSSS), 1e-4 for f16 in / f32 compute and out (HHSS).
Orientation: tinyblas(transa = 1, transb = 0), i.e. A is m rows of k with lda = k + slack — llamafile_sgemm's layout
(C = A^T B, column-major C).  alpha / beta are not part of the llamafile_sgemm boundary (alpha = 1, beta = 0 only).
"""
import numpy as np
import pytest
import torch

from llamafile_amd import ggml_types as T

pytestmark = pytest.mark.gpu

K_DIMS = [1, 2, 23, 65, 63, 64, 1024, 512, 127, 129, 128, 16]  # tester.cu:67
TOMBSTONE = np.float32(1.666)  # tester.h:51


def error_report(want, got):
    """tester.h:100-166 (diff): statistics over the outputs whose signs agree."""
    nans = int(np.isnan(want).sum() + np.isnan(got).sum())
    ok = ~(np.isnan(want) | np.isnan(got))
    w, g = want[ok].astype(np.float64), got[ok].astype(np.float64)
    same = np.signbit(w) == np.signbit(g)
    flips = int((~same).sum())
    considered = int(same.sum())
    sad = float(np.abs(w[same] - g[same]).sum() / considered) if considered else 0.0
    return nans, sad, flips


def to_type(x, t):
    if t == T.F32:
        return x.astype(np.float32)
    if t == T.F16:
        return x.astype(np.float16)
    u = x.astype(np.float32).view(np.uint32)  # bf16: nearest-even (ggml_compute_fp32_to_bf16)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def as_f64(a, t):
    if t == T.BF16:
        return (a.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    return a.astype(np.float64)


def run_case(gpu, rng, ta, tb, m, n, k, slack):
    lda, ldb, ldc = k + slack, k + slack, m + slack
    A = np.zeros((m, lda), dtype=np.float32)
    A[:, :k] = rng.uniform(-1, 1, (m, k))
    Bm = np.zeros((n, ldb), dtype=np.float32)
    Bm[:, :k] = rng.uniform(-1, 1, (n, k))
    At, Bt = to_type(A, ta), to_type(Bm, tb)
    want = as_f64(Bt[:, :k], tb) @ as_f64(At[:, :k], ta).T  # gemmref<double>
    esz = T.TYPE_SIZE[ta]
    W = gpu.upload_weights(ta, At.view(np.uint8).reshape(m, lda * esz), m, k)  # row stride lda elements
    Bd = torch.from_numpy(Bt.view(np.uint8).reshape(n, ldb * T.TYPE_SIZE[tb])).cuda()
    out = torch.full((n, ldc), float(TOMBSTONE), dtype=torch.float32, device="cuda")
    gpu.mul_mat(W, Bd, tb, n=n, out=out, ldc=ldc)
    got = out.cpu().numpy()
    if slack:
        assert (got[:, m:] == TOMBSTONE).all(), "bytes outside the m x n result were written"
    return want.astype(np.float32), got[:, :m]


@pytest.mark.parametrize("ta,tb,tol", [(T.F32, T.F32, 1e-4), (T.F16, T.F16, 1e-4), (T.BF16, T.BF16, 1e-4)],
                         ids=["SSSS", "HHSS", "BBSS"])
def test_reference_shape_sweep(gpu, ta, tb, tol):
    rng = np.random.default_rng(0x7E57)
    bad = []
    dims = K_DIMS
    for mi, m in enumerate(dims):
        for ni in range(len(dims)):
            n = dims[len(dims) - 1 - ni]
            for slack in (0, 1):
                for k in dims:
                    want, got = run_case(gpu, rng, ta, tb, m, n, k, slack)
                    nans, sad, flips = error_report(want, got)
                    if nans or sad > tol or flips >= m * n * 0.01:
                        bad.append((m, n, k, slack, nans, sad, flips))
    assert not bad, bad[:10]


@pytest.mark.parametrize("shape", [(5760, 1, 128), (128, 1, 5696), (14336, 512, 4096), (1024, 1024, 1024), (2048, 2048, 2048),
                                   (32000, 512, 4096)], ids=str)
def test_reference_try_size(gpu, shape):
    """tinyblas_test.cu:230-245 try_size: f16 operands, CHECK(1, ...) — the model-sized shapes of the reference's own
    main() (mistral 7b: 5760 x 1 x 128, 14336 x 512 x 4096, the 32000-row output matrix)."""
    m, n, k = shape
    rng = np.random.default_rng(m * 31 + n)
    want, got = run_case(gpu, rng, T.F16, T.F16, m, n, k, 0)
    nans, sad, flips = error_report(want, got)
    assert nans == 0 and sad <= 1.0 and flips < m * n * 0.01
    # the reference accepts sad <= 1 here (half outputs); f32 accumulation does far better — pin that too
    assert sad <= 1e-4, sad
