"""Fused RMS-norm x weight -> Q8_K (f-3): bytes equal to the CPU restatement of ggml's rms_norm + MUL + quantize_row_q8_K,
and the mat-mul fed with those blocks equals the mat-mul fed with the f32 result (which quantises in its own prologue)."""
import ctypes as C

import numpy as np
import pytest
import torch

from llamafile_amd import _hip, ggml_types as T, synth

pytestmark = pytest.mark.gpu


def cpu_rms_norm_mul(x, w, eps):
    y = np.empty_like(x)
    for r in range(x.shape[0]):
        s = 0.0
        for v in x[r]:  # ggml: sum += (ggml_float)(x[i] * x[i]), sequentially in double
            s += float(np.float32(v) * np.float32(v))
        mean = np.float32(s / x.shape[1])
        scale = np.float32(1.0) / np.sqrt(np.float32(mean + np.float32(eps)))
        y[r] = (x[r] * scale).astype(np.float32) * w
    return y


@pytest.mark.parametrize("n,k", [(1, 4096), (5, 1024), (3, 14336)])
def test_rms_norm_quantize_matches_cpu_and_feeds_the_gemv(gpu, oracle, n, k):
    rng = np.random.default_rng(k + n)
    x = (rng.standard_normal((n, k)) * rng.uniform(0.1, 30.0, (n, 1))).astype(np.float32)
    x[0, 256:512] = 0.0  # an all-zero block after scaling stays all zero
    w = rng.uniform(0.5, 1.5, k).astype(np.float32)
    eps = 1e-5
    xd, wd = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()
    qrow = T.row_size(T.Q8_K, k)
    yq = torch.zeros((n, qrow), dtype=torch.uint8, device="cuda")
    yf = torch.zeros((n, k), dtype=torch.float32, device="cuda")
    L = _hip.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _hip.check(L.lfamd_rms_norm_quantize(C.c_void_p(xd.data_ptr()), k * 4, C.c_void_p(wd.data_ptr()), eps, n, k, T.Q8_K,
                                         C.c_void_p(yq.data_ptr()), qrow, C.c_void_p(yf.data_ptr()), k * 4, st), "rms_norm_quantize")
    want_f = cpu_rms_norm_mul(x, w, eps)
    got_f = yf.cpu().numpy()
    assert np.array_equal(got_f.view(np.uint32), want_f.view(np.uint32)), np.abs(got_f - want_f).max()
    want_q = oracle.quantize(T.Q8_K, want_f)
    assert np.array_equal(yq.cpu().numpy(), want_q)
    # the GEMV fed with the blocks == the GEMV fed with the f32 rows (its own prologue quantises them identically)
    m = 96
    W = gpu.upload_weights(T.Q4_K, synth.random_weights(T.Q4_K, m, k, 3), m, k)
    a = gpu.mul_mat(W, yq, T.Q8_K, n=n)
    b = gpu.mul_mat(W, yf.view(torch.uint8).view(n, k * 4), T.F32, n=n)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def test_rejects_what_it_cannot_do(gpu):
    L = _hip.lib()
    z = C.c_void_p(0)
    x = torch.zeros(512, device="cuda")
    assert L.lfamd_rms_norm_quantize(C.c_void_p(x.data_ptr()), 2048, z, 1e-5, 1, 500, T.Q8_K, C.c_void_p(x.data_ptr()), 584, z, 0, None) == -2
    assert L.lfamd_rms_norm_quantize(C.c_void_p(x.data_ptr()), 2048, z, 1e-5, 1, 512, T.Q8_0, C.c_void_p(x.data_ptr()), 584, z, 0, None) == -2


@pytest.mark.parametrize("n,k", [(1, 14336), (6, 1024), (2, 4096)])
def test_swiglu_quantize_feeds_ffn_down(gpu, oracle, n, k):
    """silu(gate) * up -> Q8_K in one kernel: the f32 result within an ulp or two of the f64 formula (expf differs between
    libraries in the last bit), the blocks EXACTLY quantize_row_q8_K of the kernel's own f32 result, and ffn_down fed with the
    blocks equals ffn_down fed with that f32 result."""
    rng = np.random.default_rng(k * 3 + n)
    g = (rng.standard_normal((n, k)) * 3.0).astype(np.float32)
    u = rng.standard_normal((n, k)).astype(np.float32)
    g[0, 512:768] = 0.0  # silu(0) * up = 0: an all-zero block
    gd, ud = torch.from_numpy(g).cuda(), torch.from_numpy(u).cuda()
    qrow = T.row_size(T.Q8_K, k)
    yq = torch.zeros((n, qrow), dtype=torch.uint8, device="cuda")
    yf = torch.zeros((n, k), dtype=torch.float32, device="cuda")
    L = _hip.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _hip.check(L.lfamd_swiglu_quantize(C.c_void_p(gd.data_ptr()), k * 4, C.c_void_p(ud.data_ptr()), k * 4, n, k, T.Q8_K,
                                       C.c_void_p(yq.data_ptr()), qrow, C.c_void_p(yf.data_ptr()), k * 4, st), "swiglu_quantize")
    got_f = yf.cpu().numpy()
    want = (g.astype(np.float64) / (1.0 + np.exp(-g.astype(np.float64)))) * u.astype(np.float64)
    assert np.all(np.abs(got_f - want) <= 4e-7 * np.abs(want) + 1e-30)
    assert np.array_equal(yq.cpu().numpy(), oracle.quantize(T.Q8_K, got_f))
    m = 64
    W = gpu.upload_weights(T.Q6_K, synth.random_weights(T.Q6_K, m, k, 4), m, k)
    a = gpu.mul_mat(W, yq, T.Q8_K, n=n)
    b = gpu.mul_mat(W, yf.view(torch.uint8).view(n, k * 4), T.F32, n=n)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    z = C.c_void_p(0)
    assert L.lfamd_swiglu_quantize(C.c_void_p(gd.data_ptr()), k * 4, C.c_void_p(ud.data_ptr()), k * 4, n, k - 16, T.Q8_K,
                                   C.c_void_p(yq.data_ptr()), qrow, z, 0, None) == -2
