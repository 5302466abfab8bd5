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


def _staged_mul_mat(L, W, image, m, k, n, flags, multi=None):
    """lfamd_mul_mat (or lfamd_mul_mat_multi over `multi` = a list of PackedWeights) on a staged image; returns the outputs."""
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if multi is None:
        out = torch.empty((n, m), dtype=torch.float32, device="cuda")
        rc = L.lfamd_mul_mat(T.Q4_K, C.c_void_p(W.data.data_ptr()), m, k, _hip.TYPE_STAGED_Q8K, C.c_void_p(image.data_ptr()), 0, n,
                             C.c_void_p(out.data_ptr()), m, C.c_void_p(0), 0, flags, st)
        return rc, out
    cnt = len(multi)
    outs = [torch.empty((n, w.rows), dtype=torch.float32, device="cuda") for w in multi]
    A = (C.c_void_p * cnt)(*[w.data.data_ptr() for w in multi])
    Cs = (C.c_void_p * cnt)(*[o.data_ptr() for o in outs])
    ms = (C.c_long * cnt)(*[w.rows for w in multi])
    rc = L.lfamd_mul_mat_multi(T.Q4_K, cnt, A, ms, k, _hip.TYPE_STAGED_Q8K, C.c_void_p(image.data_ptr()), 0, n, Cs, ms, C.c_void_p(0), 0, flags, st)
    return rc, outs


@pytest.mark.parametrize("m,k,n", [(2048, 512, 512), (4096, 256, 300), (4096, 1024, 257)])
@pytest.mark.parametrize("producer", ["swiglu", "rms_norm"])
def test_producers_write_the_staged_image_of_the_int8_body(gpu, producer, m, k, n):
    """lfamd_swiglu_quantize / lfamd_rms_norm_quantize with LFAMD_TYPE_STAGED_Q8K: the Q4_K batch body on the int8 matrix cores
    reads what the producer wrote — no staging launch in front of the mat-mul — and gives the BITS of the same mat-mul on the
    producer's f32 output (which quantises and stages in its own launch); ragged token counts (the image's padding tokens are
    zero), sibling matrices on one image, and LFAMD_ERR_UNSUPPORTED where a call does not run that body."""
    L = _hip.lib()
    flags = gpu.host_variant_flags()
    assert L.lfamd_mul_mat_takes_staged(T.Q4_K, m, k, n, flags) == 1
    rng = np.random.default_rng(m + k + n)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    size = L.lfamd_staged_q8k_size(k, n)
    assert size == ((n + 127) // 128 * 128) * (k // 256) * (256 + 4 + 32)
    image = torch.full((size,), 0x5A, dtype=torch.uint8, device="cuda")
    yf = torch.zeros((n, k), dtype=torch.float32, device="cuda")
    if producer == "swiglu":
        g = torch.from_numpy((rng.standard_normal((n, k)) * 3.0).astype(np.float32)).cuda()
        u = torch.from_numpy(rng.standard_normal((n, k)).astype(np.float32)).cuda()
        _hip.check(L.lfamd_swiglu_quantize(C.c_void_p(g.data_ptr()), k * 4, C.c_void_p(u.data_ptr()), k * 4, n, k, _hip.TYPE_STAGED_Q8K,
                                           C.c_void_p(image.data_ptr()), 0, C.c_void_p(yf.data_ptr()), k * 4, st), "swiglu_quantize (staged)")
    else:
        x = torch.from_numpy((rng.standard_normal((n, k)) * rng.uniform(0.1, 30.0, (n, 1))).astype(np.float32)).cuda()
        w = torch.from_numpy(rng.uniform(0.5, 1.5, k).astype(np.float32)).cuda()
        _hip.check(L.lfamd_rms_norm_quantize(C.c_void_p(x.data_ptr()), k * 4, C.c_void_p(w.data_ptr()), 1e-5, n, k, _hip.TYPE_STAGED_Q8K,
                                             C.c_void_p(image.data_ptr()), 0, C.c_void_p(yf.data_ptr()), k * 4, st), "rms_norm_quantize (staged)")
    W = gpu.upload_weights(T.Q4_K, synth.random_weights_torch(T.Q4_K, m, k, 7).cpu().numpy(), m, k)
    rc, got = _staged_mul_mat(L, W, image, m, k, n, flags)
    assert rc == 0, L.lfamd_last_error()
    want = gpu.mul_mat(W, yf.view(torch.uint8).view(n, k * 4), T.F32, n=n)
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    # two sibling matrices on the one image
    W2 = gpu.upload_weights(T.Q4_K, synth.random_weights_torch(T.Q4_K, m, k, 8).cpu().numpy(), m, k)
    rc, outs = _staged_mul_mat(L, None, image, m, k, n, flags, multi=[W, W2])
    assert rc == 0, L.lfamd_last_error()
    assert torch.equal(outs[0].view(torch.int32), want.view(torch.int32))
    assert torch.equal(outs[1].view(torch.int32), gpu.mul_mat(W2, yf.view(torch.uint8).view(n, k * 4), T.F32, n=n).view(torch.int32))
    # calls that do not run the int8 body decline the image: a handful of rows, the exact-code flag, another type
    assert L.lfamd_mul_mat_takes_staged(T.Q4_K, 64, k, n, flags) == 0
    Ws = gpu.upload_weights(T.Q4_K, synth.random_weights(T.Q4_K, 64, k, 9), 64, k)
    rc, _ = _staged_mul_mat(L, Ws, image, 64, k, n, flags)
    assert rc == -1
    rc, _ = _staged_mul_mat(L, W, image, m, k, n, flags | _hip.FLAG_PRECISE)
    assert rc == -1
    assert L.lfamd_mul_mat_takes_staged(T.Q6_K, m, k, n, flags) == 0


def _call_multi(L, types, Ws, k, Btype, B, brb, n, flags, ws):
    cnt = len(Ws)
    outs = [torch.empty((n, w.rows), dtype=torch.float32, device="cuda") for w in Ws]
    A = (C.c_void_p * cnt)(*[w.data.data_ptr() for w in Ws])
    Cs = (C.c_void_p * cnt)(*[o.data_ptr() for o in outs])
    ms = (C.c_long * cnt)(*[w.rows for w in Ws])
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if len(set(types)) == 1:
        rc = L.lfamd_mul_mat_multi(types[0], cnt, A, ms, k, Btype, C.c_void_p(B.data_ptr()), brb, n, Cs, ms, C.c_void_p(ws.data_ptr()), ws.numel(), flags, st)
    else:
        ts = (C.c_int * cnt)(*types)
        rc = L.lfamd_mul_mat_multi_types(cnt, ts, A, ms, k, Btype, C.c_void_p(B.data_ptr()), brb, n, Cs, ms, C.c_void_p(ws.data_ptr()), ws.numel(), flags, st)
    return rc, outs


@pytest.mark.parametrize("k,n", [(512, 512), (1024, 300)])
@pytest.mark.parametrize("producer", ["rms_norm", "swiglu"])
def test_producers_write_the_scaled_image_of_the_f16_bodies(gpu, producer, k, n):
    """LFAMD_TYPE_STAGED_SCALED: the image prep_scaled_kernel would write (f16 operands with a per-token power-of-two normalisation,
    mins operand, 2^e per token) comes straight from the fused producer; single matrices (Q4_K on a large grid, Q6_K), sibling
    matrices in one launch (ffn_gate + ffn_up) and siblings of two types (attn_q/k Q4_K + attn_v Q6_K) read it and give the BITS of
    the same calls on the producer's f32 output."""
    L = _hip.lib()
    flags = gpu.host_variant_flags()
    rng = np.random.default_rng(k * 7 + n)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    size = L.lfamd_staged_scaled_size(k, n)
    image = torch.full((size,), 0x5A, dtype=torch.uint8, device="cuda")
    yf = torch.zeros((n, k), dtype=torch.float32, device="cuda")
    if producer == "swiglu":
        g = torch.from_numpy((rng.standard_normal((n, k)) * 3.0).astype(np.float32)).cuda()
        u = torch.from_numpy((rng.standard_normal((n, k)) * rng.uniform(1e-3, 50.0, (n, 1))).astype(np.float32)).cuda()
        g[1, :] = 0.0  # an all-zero row: no normalisation
        _hip.check(L.lfamd_swiglu_quantize(C.c_void_p(g.data_ptr()), k * 4, C.c_void_p(u.data_ptr()), k * 4, n, k, _hip.TYPE_STAGED_SCALED,
                                           C.c_void_p(image.data_ptr()), 0, C.c_void_p(yf.data_ptr()), k * 4, st), "swiglu_quantize (scaled image)")
    else:
        x = torch.from_numpy((rng.standard_normal((n, k)) * rng.uniform(1e-3, 50.0, (n, 1))).astype(np.float32)).cuda()
        w = torch.from_numpy(rng.uniform(0.5, 1.5, k).astype(np.float32)).cuda()
        _hip.check(L.lfamd_rms_norm_quantize(C.c_void_p(x.data_ptr()), k * 4, C.c_void_p(w.data_ptr()), 1e-5, n, k, _hip.TYPE_STAGED_SCALED,
                                             C.c_void_p(image.data_ptr()), 0, C.c_void_p(yf.data_ptr()), k * 4, st), "rms_norm_quantize (scaled image)")
    yb = yf.view(torch.uint8).view(n, k * 4)

    def W(t, m, seed):
        return gpu.upload_weights(t, synth.random_weights_torch(t, m, k, seed).cpu().numpy(), m, k)

    # single matrices: a Q4_K grid that fills the chip (row-split body), a Q6_K matrix (loader-wave body)
    for t, m in ((T.Q4_K, 14336), (T.Q6_K, 4096), (T.Q5_K, 1000)):
        assert L.lfamd_mul_mat_takes_staged_scaled(t, m, k, n, flags) == 1, (T.NAMES[t], m)
        Wt = W(t, m, 11)
        ws = torch.empty(max(16, L.lfamd_mul_mat_workspace(t, m, k, n)), dtype=torch.uint8, device="cuda")
        out = torch.empty((n, m), dtype=torch.float32, device="cuda")
        rc = L.lfamd_mul_mat(t, C.c_void_p(Wt.data.data_ptr()), m, k, _hip.TYPE_STAGED_SCALED, C.c_void_p(image.data_ptr()), 0, n,
                             C.c_void_p(out.data_ptr()), m, C.c_void_p(ws.data_ptr()), ws.numel(), flags, st)
        assert rc == 0, L.lfamd_last_error()
        want = gpu.mul_mat(Wt, yb, T.F32, n=n)
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int32), want.view(torch.int32)), (T.NAMES[t], m)
    # ffn_gate + ffn_up: one launch over both
    Ws = [W(T.Q4_K, 14336, 21), W(T.Q4_K, 14336, 22)]
    ws = torch.empty(max(L.lfamd_mul_mat_workspace(T.Q4_K, 14336, k, n), 16), dtype=torch.uint8, device="cuda")
    rc, got = _call_multi(L, [T.Q4_K, T.Q4_K], Ws, k, _hip.TYPE_STAGED_SCALED, image, 0, n, flags, ws)
    assert rc == 0, L.lfamd_last_error()
    rc, want = _call_multi(L, [T.Q4_K, T.Q4_K], Ws, k, T.F32, yb, k * 4, n, flags, ws)
    assert rc == 0
    torch.cuda.synchronize()
    for a, b in zip(got, want):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    # attn_q / attn_k (Q4_K) + attn_v (Q6_K): one staging for both types
    types = [T.Q4_K, T.Q4_K, T.Q6_K]
    Ws = [W(T.Q4_K, 4096, 31), W(T.Q4_K, 1024, 32), W(T.Q6_K, 1024, 33)]
    ws = torch.empty(max(L.lfamd_mul_mat_workspace(T.Q4_K, 4096, k, n), L.lfamd_mul_mat_workspace(T.Q6_K, 1024, k, n), 16), dtype=torch.uint8, device="cuda")
    rc, got = _call_multi(L, types, Ws, k, _hip.TYPE_STAGED_SCALED, image, 0, n, flags, ws)
    assert rc == 0, L.lfamd_last_error()
    rc, want = _call_multi(L, types, Ws, k, T.F32, yb, k * 4, n, flags, ws)
    assert rc == 0
    torch.cuda.synchronize()
    for a, b in zip(got, want):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    # a call that runs another body declines the image
    assert L.lfamd_mul_mat_takes_staged_scaled(T.Q4_K, 4096, k, n, flags | _hip.FLAG_PRECISE) == 0
    assert L.lfamd_mul_mat_takes_staged_scaled(T.Q8_0, 4096, k, n, flags) == 0
