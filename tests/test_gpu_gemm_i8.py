"""The Q4_K prefill body on the int8 matrix cores (csrc/gemm_i8.hip) against the oracle: exact integer sub-block dots, the
reference's f32 scale arithmetic (iqk_mul_mat.inc:601-643) — 2e-6 normwise AND element-wise, f32 and pre-quantised input,
ragged rows / tokens, ldc > m, several matrices in one launch."""
import ctypes as C

import numpy as np
import pytest
import torch

from llamafile_amd import _hip, ggml_types as T, synth
from helpers import elem_err, rel_err

pytestmark = pytest.mark.gpu

# (m, n, k): at least 128 tiles of 128 x 64 and at most 256 of 128 x 128 (two rounds of its tiles), so the default route is the int8 body
SHAPES = [(2048, 512, 512), (4096, 256, 256), (1000, 1100, 768), (16384 - 32 + 5, 70, 256), (4096 + 5, 257, 1024), (6144 + 9, 512, 512),
          (8192, 500, 256)]


def _oracle_rows(oracle, A, B, m, n, k, rows, cols):
    """oracle on a sample: `rows` of the weight matrix x `cols` of the tokens (the full product takes minutes at these sizes)."""
    rb = T.row_size(T.Q4_K, k)
    As = np.ascontiguousarray(A.reshape(m, rb)[rows])
    Bs = np.ascontiguousarray(B.reshape(n, -1)[cols])
    ok, G = oracle.sgemm(T.Q4_K, As, T.Q8_K, Bs, len(rows), len(cols), k, nth=4)
    assert ok == 1
    return G


@pytest.mark.parametrize("shape", SHAPES, ids=str)
@pytest.mark.parametrize("f32in", [False, True], ids=["q8k", "f32"])
def test_int8_body_vs_oracle(gpu, oracle, shape, f32in):
    m, n, k = shape
    A = synth.random_weights(T.Q4_K, m, k, 41)
    x = synth.random_activations(n, k, 42)
    B = synth.quantize_activations(T.Q8_K, x)
    assert _hip.lib().lfamd_mul_mat_is_exact(T.Q4_K, m, k, n, gpu.host_variant_flags()) == 1  # (the default route: the int8 body)
    W = gpu.upload_weights(T.Q4_K, A, m, k)
    Bd = torch.from_numpy(x).cuda().view(torch.uint8).view(n, k * 4) if f32in else torch.from_numpy(B).cuda()
    Cd = gpu.mul_mat(W, Bd, T.F32 if f32in else T.Q8_K, n=n)
    Cp = gpu.mul_mat(W, Bd, T.F32 if f32in else T.Q8_K, n=n, flags=gpu.host_variant_flags() | _hip.FLAG_PRECISE)  # exact-code f16 body
    torch.cuda.synchronize()
    Cn, Cpn = Cd.cpu().numpy(), Cp.cpu().numpy()
    assert not np.isnan(Cn).any()
    # the whole product against the exact-code f16 body (same integers, same scale arithmetic up to f32 summation order)
    assert rel_err(Cn, Cpn) <= 2e-6, rel_err(Cn, Cpn)
    # a sample against the oracle: first / last rows of every row tile class, first / last tokens
    rng = np.random.default_rng(5)
    rows = np.unique(np.concatenate([np.arange(0, min(m, 40)), np.arange(max(0, m - 40), m), rng.integers(0, m, 48)]))
    cols = np.unique(np.concatenate([np.arange(0, min(n, 8)), np.arange(max(0, n - 8), n), rng.integers(0, n, 24)]))
    G = _oracle_rows(oracle, A, B, m, n, k, rows, cols)
    Cs = Cn[np.ix_(cols, rows)]
    assert rel_err(Cs, G) <= 2e-6, rel_err(Cs, G)
    frac, worst = elem_err(Cs, G, rtol=1e-5)
    assert frac == 0.0, (frac, worst)


def test_int8_body_respects_ldc_and_leaves_the_rest_alone(gpu):
    m, n, k, ldc = 2048, 512, 256, 2048 + 64
    A = synth.random_weights(T.Q4_K, m, k, 51)
    x = synth.random_activations(n, k, 52)
    W = gpu.upload_weights(T.Q4_K, A, m, k)
    Bd = torch.from_numpy(x).cuda().view(torch.uint8).view(n, k * 4)
    ref = gpu.mul_mat(W, Bd, T.F32, n=n)
    out = torch.full((n, ldc), 7.0, device="cuda")
    L = _hip.lib()
    ws = torch.empty(gpu.workspace_bytes(T.Q4_K, m, k, n), dtype=torch.uint8, device="cuda")
    rc = L.lfamd_mul_mat(T.Q4_K, C.c_void_p(W.data.data_ptr()), m, k, T.F32, C.c_void_p(Bd.data_ptr()), k * 4, n, C.c_void_p(out.data_ptr()),
                         ldc, C.c_void_p(ws.data_ptr()), ws.numel(), gpu.host_variant_flags(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, L.lfamd_last_error()
    torch.cuda.synchronize()
    assert torch.equal(out[:, :m], ref) and bool((out[:, m:] == 7.0).all())


def test_int8_body_serves_sibling_matrices_in_one_launch(gpu, oracle):
    """lfamd_mul_mat_multi on Q4_K siblings whose row blocks TOGETHER make a grid the int8 body takes (attn_q/k/v of an all-Q4_K
    layer): one staging, one launch, exact integer dots — every matrix 2e-6 of the oracle on a sample, although the small ones alone
    would run a scaled-operand body."""
    k, n = 512, 512
    ms = [4096, 1024, 1024]
    L = _hip.lib()
    assert L.lfamd_mul_mat_is_exact(T.Q4_K, 1024, k, n, gpu.host_variant_flags()) == 0  # (alone: the loader-wave body)
    As = [synth.random_weights(T.Q4_K, m, k, 61 + i) for i, m in enumerate(ms)]
    Ws = [gpu.upload_weights(T.Q4_K, A, m, k) for A, m in zip(As, ms)]
    x = synth.random_activations(n, k, 62)
    B = synth.quantize_activations(T.Q8_K, x)
    fused = gpu.mul_mat_multi(Ws, torch.from_numpy(x).cuda().view(torch.uint8).view(n, k * 4), T.F32, n=n)
    rng = np.random.default_rng(6)
    for A, m, f in zip(As, ms, fused):
        rows = np.unique(np.concatenate([np.arange(0, 24), np.arange(m - 24, m), rng.integers(0, m, 32)]))
        cols = np.unique(np.concatenate([np.arange(0, 8), np.arange(n - 8, n), rng.integers(0, n, 16)]))
        G = _oracle_rows(oracle, A, B, m, n, k, rows, cols)
        Cs = f.cpu().numpy()[np.ix_(cols, rows)]
        assert rel_err(Cs, G) <= 2e-6, (m, rel_err(Cs, G))
        frac, worst = elem_err(Cs, G, rtol=1e-5)
        assert frac == 0.0, (m, frac, worst)
