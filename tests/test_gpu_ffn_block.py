"""lfamd_ffn_block (f-3): the decode feed-forward block as one launch, against
  oracle(ffn_gate), oracle(ffn_up)  ->  f64 swiglu  ->  Q8_K (oracle quantiser)  ->  oracle(ffn_down) (+ residual)
and against the separate HIP calls it replaces (gate + up GEMV, lfamd_swiglu_quantize, ffn_down GEMV), at the shapes of
Llama-3-8B (k = 4096, n_ff = 14336; ffn_down in Q4_K and in Q6_K as in a Q4_K_M file) and at ragged small ones."""
import ctypes as C

import numpy as np
import pytest
import torch

from llamafile_amd import _hip, ggml_types as T, synth
from helpers import rel_err

pytestmark = pytest.mark.gpu


def _oracle_ffn(oracle, Ag, Au, Ad, td, x, k, n_ff, m, residual):
    bq = oracle.quantize(T.Q8_K, x)
    ok, g = oracle.sgemm(T.Q4_K, Ag, T.Q8_K, bq, n_ff, 1, k, nth=8)
    assert ok == 1
    ok, u = oracle.sgemm(T.Q4_K, Au, T.Q8_K, bq, n_ff, 1, k, nth=8)
    assert ok == 1
    g64, u64 = g.astype(np.float64), u.astype(np.float64)
    h = ((g64 / (1.0 + np.exp(-g64))) * u64).astype(np.float32)  # f64 swiglu, rounded once
    hq = oracle.quantize(T.Q8_K, h)
    ok, y = oracle.sgemm(td, Ad, T.Q8_K, hq, m, 1, n_ff, nth=8)
    assert ok == 1
    return (y + residual if residual is not None else y), h


@pytest.mark.parametrize("td", [T.Q4_K, T.Q6_K], ids=lambda t: T.NAMES[t])
@pytest.mark.parametrize("shape", [(4096, 14336, 4096), (1024, 2560, 200), (512, 768, 4096 + 32)], ids=str)
@pytest.mark.parametrize("with_residual", [False, True], ids=["plain", "residual"])
def test_ffn_block_vs_oracle_and_separate_calls(gpu, oracle, td, shape, with_residual):
    k, n_ff, m = shape
    Ag, Au = synth.random_weights(T.Q4_K, n_ff, k, 11), synth.random_weights(T.Q4_K, n_ff, k, 12)
    Ad = synth.random_weights(td, m, n_ff, 13)
    x = synth.random_activations(1, k, 14)
    res = synth.random_activations(1, m, 15) if with_residual else None
    want, _ = _oracle_ffn(oracle, Ag, Au, Ad, td, x, k, n_ff, m, res)

    Wg, Wu = gpu.upload_weights(T.Q4_K, Ag, n_ff, k), gpu.upload_weights(T.Q4_K, Au, n_ff, k)
    Wd = gpu.upload_weights(td, Ad, m, n_ff)
    xd = torch.from_numpy(x).cuda()
    rd = torch.from_numpy(res).cuda() if with_residual else None
    out = gpu.ffn_block(Wg, Wu, Wd, xd, residual=rd)
    out2 = gpu.ffn_block(Wg, Wu, Wd, xd, residual=rd)  # (the barrier state returns to zero arrivals after every launch)
    torch.cuda.synchronize()
    assert _hip.lib().lfamd_ffn_block_check() == 0
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    assert torch.equal(out.view(torch.int32), out2.view(torch.int32)), "a rerun differs"
    # h is quantised again between the two mat-muls: a last-bit difference of h (f32 expf on the device, f64 here) moves a
    # Q8_K code by one step wherever the scaled value sits on a rounding boundary (1/127 of that term; with random weights
    # |h| spans five decades).  Measured 1e-5 .. 8e-4 normwise; the bound is the north star's (logits within 1e-3 relative).
    # The sharp check is the one below: the separate HIP calls, each of which is oracle-tested on its own.
    assert rel_err(got, want) <= 1e-3, rel_err(got, want)

    # the separate calls it replaces
    xb = xd.view(torch.uint8).view(1, k * 4)
    g, u = gpu.mul_mat_multi([Wg, Wu], xb, T.F32, n=1)
    L = _hip.lib()
    qrow = T.row_size(T.Q8_K, n_ff)
    hq = torch.zeros((1, qrow), dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _hip.check(L.lfamd_swiglu_quantize(C.c_void_p(g.data_ptr()), n_ff * 4, C.c_void_p(u.data_ptr()), n_ff * 4, 1, n_ff, T.Q8_K,
                                       C.c_void_p(hq.data_ptr()), qrow, None, 0, st), "swiglu_quantize")
    y = gpu.mul_mat(Wd, hq, T.Q8_K, n=1)
    if with_residual:
        y = y + rd
    torch.cuda.synchronize()
    # same integer dots, same quantiser; the f32 sums over a row's super-blocks follow the wave layout (14 or 15 computing
    # waves here, 8 or 16 in the GEMV): last-bit differences only
    assert rel_err(got, y.cpu().numpy()) <= 2e-6


def test_ffn_block_declines_what_it_does_not_cover(gpu):
    k, n_ff, m = 512, 768, 64
    Wg = gpu.upload_weights(T.Q6_K, synth.random_weights(T.Q6_K, n_ff, k, 1), n_ff, k)
    Wd = gpu.upload_weights(T.Q4_K, synth.random_weights(T.Q4_K, m, n_ff, 2), m, n_ff)
    x = torch.zeros((1, k), device="cuda")
    with pytest.raises(_hip.LfamdError):
        gpu.ffn_block(Wg, Wg, Wd, x)  # Q6_K gate / up: the separate calls
    L = _hip.lib()
    assert L.lfamd_ffn_block_workspace(14336) == 2 * 14336 * 4
