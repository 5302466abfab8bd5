"""Decode GEMV (n = 1..8): every quantised type through every kernel variant the dispatcher can pick — selected by the
row depth and the number of half-tiles: 8 waves x 2 super-blocks (<= one half-tile per CU, <= 32 super-blocks), 16 waves
x 1 (more half-tiles, <= 16 super-blocks), 16 waves x 2 (deeper rows), the multi-column body (n > 1) — with f32
activations (quantised in the kernel: one block per wave, two blocks per pass) and pre-quantised ones, against the oracle.
The f32 and the pre-quantised launch must agree bit for bit (same integer dots, same f32 order)."""
import os

import numpy as np
import pytest
import torch

from llamafile_amd import ggml_types as T, synth
from helpers import rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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


# ---- small batches on the matrix cores (csrc/gemm_sb.hip): SB_MIN .. 32 tokens of Q4_K / Q5_K / Q6_K
# (which body runs: api.hip use_gemm_sb — deep rows from 3 tokens; one row tile per CU from 5; up to four tiles per CU 8..24 tokens,
#  where a work-group walks several tiles on two alternating weight register sets: m = 9000 -> 2 tiles, m = 20000 -> 3)
#  — on the 16-wave body with the codes in LDS up to 16 tokens; (8500, 8192): its K-halves form)
SB_SHAPES = [(48, 768), (33, 256), (100, 4352), (4096, 4096), (24, 14336), (1000, 2048), (9000, 512), (20000, 256), (8500, 8192)]


@pytest.mark.parametrize("n", [3, 5, 8, 12, 17, 32])
@pytest.mark.parametrize("m,k", SB_SHAPES, ids=lambda v: str(v))
@pytest.mark.parametrize("t", [T.Q4_K, T.Q5_K, T.Q6_K], ids=lambda t: T.NAMES[t])
def test_small_batch_mfma_vs_oracle(gpu, oracle, t, m, k, n):
    """A handful of tokens: exact integer codes as f16 fragments, one MFMA tile of 32 token slots, two work-groups (K halves) adding
    into the zeroed result.  <= 2e-6 of the oracle; f32 and pre-quantised activations give the same bits; a rerun gives the same
    bits (two addends commute); rows and tokens beside the result are left alone."""
    raw = synth.random_weights_torch(t, m, k, 500 + t + k % 89).cpu().numpy()
    x = synth.random_activations(n, k, 23 + k % 11)
    x[n - 1, :256] = 0.0  # an all-zero block
    bt = T.VEC_DOT[t]
    Bq = synth.quantize_activations(bt, x)
    W = gpu.upload_weights(t, raw, m, k)
    xd = torch.from_numpy(x).cuda().view(torch.uint8).view(n, k * 4)
    c_f32 = gpu.mul_mat(W, xd, T.F32, n=n).cpu().numpy()
    c_again = gpu.mul_mat(W, xd, T.F32, n=n).cpu().numpy()
    c_q = gpu.mul_mat(W, torch.from_numpy(Bq).cuda(), bt, n=n).cpu().numpy()
    assert np.array_equal(c_f32.view(np.uint32), c_again.view(np.uint32))
    assert np.array_equal(c_f32.view(np.uint32), c_q.view(np.uint32))
    rows = np.arange(m) if m <= 256 else np.unique(np.concatenate([np.arange(0, m, 61), np.arange(40), np.arange(m - 40, m)]))
    ok, G = oracle.sgemm(t, np.ascontiguousarray(raw[rows]), bt, Bq, len(rows), n, k, nth=4)
    assert ok == 1
    tiles_per_cu = ((m + 31) // 32 + 255) // 256
    if t == T.Q4_K and n <= 8:  # the int8 body (api.hip: use_gemm_sb)
        small = n >= 2 if (k > 8192 or tiles_per_cu > 4) else n >= 3 if tiles_per_cu > 1 else n >= 4
    else:
        small = (n >= 3) if k > 8192 else (n >= 5) if tiles_per_cu <= 1 else (tiles_per_cu <= 4 and t != T.Q6_K and 6 <= n <= 24)
    # (a batch of more than 8 tokens that the dispatcher keeps on the 128-token GEMM tiles runs the scaled-operand body: 1e-3)
    tol = 2e-6 if small or n <= 8 else 1e-3
    assert rel_err(c_q[:, rows], G) <= tol, (T.NAMES[t], m, k, n, rel_err(c_q[:, rows], G))
    # every column equals the single-token GEMV of that token as well (an independent kernel)
    one = gpu.mul_mat(W, xd[n // 2:n // 2 + 1].contiguous(), T.F32, n=1).cpu().numpy()
    assert rel_err(c_f32[n // 2:n // 2 + 1], one) <= tol


def test_small_batch_respects_the_result_stride(gpu, oracle):
    """ldc > m: the staging pass zeroes and the kernel adds into rows of the result only (the gap keeps its bytes)."""
    import ctypes as C
    from llamafile_amd import _hip, sgemm
    t, m, k, n, ldc = T.Q4_K, 70, 1024, 6, 96
    raw = synth.random_weights_torch(t, m, k, 77).cpu().numpy()
    x = synth.random_activations(n, k, 78)
    W = gpu.upload_weights(t, raw, m, k)
    xd = torch.from_numpy(x).cuda()
    out = torch.full((n, ldc), 7.0, device="cuda")
    ws = torch.empty(max(16, sgemm.workspace_bytes(t, m, k, n)), dtype=torch.uint8, device="cuda")
    L = _hip.lib()
    rc = L.lfamd_mul_mat(t, C.c_void_p(W.data.data_ptr()), m, k, T.F32, C.c_void_p(xd.data_ptr()), k * 4, n, C.c_void_p(out.data_ptr()), ldc,
                         C.c_void_p(ws.data_ptr()), ws.numel(), 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, L.lfamd_last_error()
    got = out.cpu().numpy()
    assert (got[:, m:] == 7.0).all()
    ok, G = oracle.sgemm(t, raw, T.Q8_K, synth.quantize_activations(T.Q8_K, x), m, n, k, nth=4)
    assert ok == 1 and rel_err(got[:, :m], G) <= 2e-6


@pytest.mark.gpu
def test_32_row_items_give_the_same_bits(gpu, tmp_path):
    """gemv_kq_body1<..., PAIR>: long walks take both half-tiles of a 32-row tile as one item (one barrier per tile).  The
    launch rule needs >= 16 half-tiles per work-group (output.weight) and a type with a long dot (Q6_K, Q2_K, Q3_K, IQ4_XS:
    gemv_impl.h kq_pair_items); LFAMD_GEMV_PAIR_MIN=1 forces the form on every 16-wave / one-super-block launch of those types,
    here 8192 x 4096 and 8200 x 2048 (ragged last tile) and a two-matrix launch: every result must be the bytes the 16-row
    items give (Q4_K rides along unchanged; the env is read once per process: two child processes)."""
    import subprocess
    import sys
    script = tmp_path / "pair_items.py"
    script.write_text(
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from llamafile_amd import ggml_types as T, sgemm, synth\n"
        "sgemm.init(0)\n"
        "outs = []\n"
        "for t, m, k in ((T.Q6_K, 8192, 4096), (T.Q2_K, 8192, 4096), (T.Q3_K, 8200, 2048), (T.IQ4_XS, 8200, 2048), (T.Q4_K, 8192, 4096)):\n"
        "    W = sgemm.upload_weights(t, synth.random_weights(t, m, k, 3), m, k)\n"
        "    x = torch.from_numpy(synth.random_activations(1, k, 4)).cuda()\n"
        "    outs.append(sgemm.mul_mat(W, x.view(torch.uint8).view(1, k * 4), T.F32, n=1).cpu().numpy())\n"
        "Ws = [sgemm.upload_weights(T.Q6_K, synth.random_weights(T.Q6_K, m, 4096, 5 + i), m, 4096) for i, m in enumerate((6144, 5120))]\n"
        "x = torch.from_numpy(synth.random_activations(1, 4096, 6)).cuda()\n"
        "outs += [o.cpu().numpy() for o in sgemm.mul_mat_multi(Ws, x.view(torch.uint8).view(1, 4096 * 4), T.F32, n=1)]\n"
        "np.savez(sys.argv[1], *outs)\n")
    res = []
    for name, env in (("items16", {"LFAMD_GEMV_PAIR_MIN": "0"}), ("items32", {"LFAMD_GEMV_PAIR_MIN": "1"})):
        out = tmp_path / (name + ".npz")
        r = subprocess.run([sys.executable, str(script), str(out)], capture_output=True, text=True, timeout=600, env={**os.environ, **env})
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(np.load(out))
    assert len(res[0].files) == 7
    for f in res[0].files:
        assert np.array_equal(res[0][f].view(np.uint32), res[1][f].view(np.uint32)), f
        assert np.isfinite(res[0][f]).all() and np.abs(res[0][f]).max() > 0
