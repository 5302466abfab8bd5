"""Decode GEMV (n = 1..8): every quantised type through every kernel variant the dispatcher can pick — selected by the
row depth and the number of half-tiles: 8 waves x 2 super-blocks (<= one half-tile per CU, <= 32 super-blocks), 16 waves
x 1 (more half-tiles, <= 16 super-blocks), 16 waves x 2 (deeper rows), the multi-column body (n > 1) — with f32
activations (quantised in the kernel: one block per wave, two blocks per pass) and pre-quantised ones, against the oracle.
The f32 and the pre-quantised launch must agree bit for bit (same integer dots, same f32 order)."""
import numpy as np
import pytest
import torch

from llamafile_amd import ggml_types as T, synth
from helpers import rel_err

pytestmark = pytest.mark.gpu

TYPES = [T.Q4_K, T.Q5_K, T.Q6_K, T.Q2_K, T.Q3_K, T.IQ4_XS, T.Q4_0, T.Q4_1, T.Q5_0, T.Q5_1]
# (m, k): k/256 = 3, 17, 32, 40, 56 super-blocks; m = 8208 rows = 513 half-tiles (> one per CU: the 16-wave forms)
SHAPES = [(48, 768), (40, 4352), (33, 8192), (8208, 1024), (8208, 4352), (64, 10240), (24, 14336)]


@pytest.mark.parametrize("n", [1, 3])
@pytest.mark.parametrize("m,k", SHAPES, ids=lambda v: str(v))
@pytest.mark.parametrize("t", TYPES, ids=lambda t: T.NAMES[t])
def test_decode_variant_vs_oracle(gpu, oracle, t, m, k, n):
    raw = synth.random_weights_torch(t, m, k, 300 + t + k % 97).cpu().numpy()
    x = synth.random_activations(n, k, 17 + k % 13)
    x[0, 256:512] = 0.0  # an all-zero block
    bt = T.VEC_DOT[t]
    Bq = synth.quantize_activations(bt, x)
    W = gpu.upload_weights(t, raw, m, k)
    c_f32 = gpu.mul_mat(W, torch.from_numpy(x).cuda().view(torch.uint8).view(n, k * 4), T.F32, n=n).cpu().numpy()
    c_q = gpu.mul_mat(W, torch.from_numpy(Bq).cuda(), bt, n=n).cpu().numpy()
    assert np.array_equal(c_f32.view(np.uint32), c_q.view(np.uint32))
    rows = np.arange(m) if m <= 256 else np.unique(np.concatenate([np.arange(0, m, 61), np.arange(40), np.arange(m - 40, m)]))
    ok, G = oracle.sgemm(t, np.ascontiguousarray(raw[rows]), bt, Bq, len(rows), n, k, nth=4)
    assert ok == 1
    assert rel_err(c_q[:, rows], G) <= 1e-5, (T.NAMES[t], m, k, n, rel_err(c_q[:, rows], G))
