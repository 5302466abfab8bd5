"""GPU parity at BASELINE.json's full sizes (Llama-3-8B shapes, 512-token prefill, batch-1 decode).

The decode GEMVs are compared with the oracle directly (one activation row is cheap for the CPU restatement).  The
512-token GEMMs are tied to them through properties that hold at any size: a GEMM column equals the GEMV of the same
activation row (same reference arithmetic), permuting weight rows / activation rows permutes the outputs bit for bit,
padding tokens do not leak into the live columns, and a rerun is bit-identical.
"""
import numpy as np
import pytest
import torch

from llamafile_amd import ggml_types as T
from helpers import rel_err

pytestmark = pytest.mark.gpu

# (weight type, m, k): attn_output / ffn_gate+up / ffn_down of Llama-3-8B Q4_K_M, the Q6_K ffn_down of its more-bits layers
SHAPES = [(T.Q4_K, 4096, 4096), (T.Q4_K, 14336, 4096), (T.Q4_K, 4096, 14336), (T.Q6_K, 4096, 14336)]
IDS = [f"{T.NAMES[t]}-{m}x{k}" for t, m, k in SHAPES]
PREFILL = 512


def _weights(gpu, t, m, k, seed):
    from llamafile_amd import synth
    raw = synth.random_weights_torch(t, m, k, seed)  # uint8 [m, row_bytes] on the device
    return raw, gpu.upload_weights(t, raw, m, k)


def _bits(x):
    return x.contiguous().view(torch.int32)


@pytest.mark.parametrize("t,m,k", SHAPES, ids=IDS)
def test_decode_gemv_full_size_vs_oracle(gpu, oracle, t, m, k):
    """Batch-1 GEMV at the model's real shapes against the CPU restatement (exact integer dots, f32 scales: 2e-6)."""
    from llamafile_amd import synth
    raw, W = _weights(gpu, t, m, k, 11)
    x = synth.random_activations(1, k, 12)
    B = synth.quantize_activations(T.Q8_K, x)
    ok, G = oracle.sgemm(t, raw.cpu().numpy(), T.Q8_K, B, m, 1, k, nth=8)
    assert ok == 1
    c_q = gpu.mul_mat(W, torch.from_numpy(B).cuda(), T.Q8_K)
    c_f = gpu.mul_mat(W, torch.from_numpy(x).cuda().view(torch.uint8).view(1, k * 4), T.F32)
    torch.cuda.synchronize()
    assert rel_err(c_q.cpu().numpy(), G) <= 2e-6
    assert torch.equal(_bits(c_q), _bits(c_f))  # in-kernel quantisation == quantize_row_q8_K


@pytest.mark.parametrize("t,m,k", SHAPES, ids=IDS)
def test_prefill_gemm_full_size_properties(gpu, t, m, k):
    from llamafile_amd import _hip, synth
    raw, W = _weights(gpu, t, m, k, 21)
    x = torch.from_numpy(synth.random_activations(PREFILL, k, 22)).cuda()
    xb = x.view(torch.uint8).view(PREFILL, k * 4)
    base = gpu.host_variant_flags()
    C = gpu.mul_mat(W, xb, T.F32, flags=base)
    C2 = gpu.mul_mat(W, xb, T.F32, flags=base)
    torch.cuda.synchronize()
    assert torch.isfinite(C).all()
    assert torch.equal(_bits(C), _bits(C2)), "rerun differs"

    # a column of the batch == the decode GEMV of that activation row (north-star tolerance for the MFMA operand
    # rounding; the exact-code GEMM bodies agree with the GEMV to the f32 summation order)
    cols = [0, 1, 63, 64, 127, 128, 255, 300, 511]
    scale = float(C.abs().max())
    for j in cols:
        g = gpu.mul_mat(W, xb[j:j + 1].contiguous(), T.F32, flags=base)
        assert float((C[j] - g[0]).abs().max()) / scale <= 1e-3, j
    if t != T.Q6_K:
        Ce = gpu.mul_mat(W, xb, T.F32, flags=base | _hip.FLAG_PRECISE)
        for j in cols:
            g = gpu.mul_mat(W, xb[j:j + 1].contiguous(), T.F32, flags=base)
            assert float((Ce[j] - g[0]).abs().max()) / scale <= 2e-6, j
        assert float((Ce - C).abs().max()) / scale <= 1e-3

    # permuting the activation rows permutes the output columns bit for bit (no dependence on the tile position)
    perm = torch.randperm(PREFILL, generator=torch.Generator().manual_seed(5)).cuda()
    Cp = gpu.mul_mat(W, xb[perm].contiguous(), T.F32, flags=base)
    assert torch.equal(_bits(Cp), _bits(C[perm]))

    # permuting the weight rows (raw GGUF rows, re-packed) permutes the output rows bit for bit
    rperm = torch.randperm(m, generator=torch.Generator().manual_seed(6)).cuda()
    Wp = gpu.upload_weights(t, raw[rperm].contiguous(), m, k)
    Cr = gpu.mul_mat(Wp, xb, T.F32, flags=base)
    assert torch.equal(_bits(Cr), _bits(C[:, rperm]))

    # ragged batch: 500 of the 512 rows give the same 500 columns (padding tokens are zero and never stored)
    Cs = gpu.mul_mat(W, xb[:500].contiguous(), T.F32, flags=base)
    assert Cs.shape[0] == 500 and torch.equal(_bits(Cs), _bits(C[:500]))


def test_gate_up_fused_launch_tail_split(gpu):
    """ffn_gate + ffn_up at 512 tokens as ONE call: 896 tiles of 128 x 128 = three full rounds + 128, whose last round runs
    as 128 x 64 tiles (the second matrix is split at a row-block boundary).  Bit-identical to the two separate calls, which
    run 128 x 128 tiles only (448 tiles each: no such tail) — i.e. the tile shape does not change a single bit."""
    from llamafile_amd import synth
    t, m, k = T.Q4_K, 14336, 4096
    Ws = [gpu.upload_weights(t, synth.random_weights_torch(t, m, k, 31 + i), m, k) for i in range(2)]
    x = torch.from_numpy(synth.random_activations(PREFILL, k, 32)).cuda()
    xb = x.view(torch.uint8).view(PREFILL, k * 4)
    fused = gpu.mul_mat_multi(Ws, xb, T.F32, n=PREFILL)
    for W, f in zip(Ws, fused):
        sep = gpu.mul_mat(W, xb, T.F32)
        assert torch.isfinite(f).all()
        assert torch.equal(_bits(f), _bits(sep))


@pytest.mark.parametrize("ta", [T.Q4_K, T.Q5_K], ids=lambda t: T.NAMES[t])
def test_qkv_two_types_one_gemm_launch(gpu, ta):
    """attn_q / attn_k (Q4_K or Q5_K) and attn_v (Q6_K) at 512 tokens through lfamd_mul_mat_multi_types: 192 tiles of
    128 x 128 in ONE launch whose work-groups run their own type's body.  Bit-identical to the three separate calls (which run
    128 x 64 tiles), in the caller's node order q, v, k as well."""
    from llamafile_amd import synth
    k = 4096
    specs = [(ta, 4096), (T.Q6_K, 1024), (ta, 1024)]
    Ws = [gpu.upload_weights(t, synth.random_weights_torch(t, m, k, 41 + i), m, k) for i, (t, m) in enumerate(specs)]
    x = torch.from_numpy(synth.random_activations(PREFILL, k, 42)).cuda()
    xb = x.view(torch.uint8).view(PREFILL, k * 4)
    fused = gpu.mul_mat_multi(Ws, xb, T.F32, n=PREFILL)
    for W, f in zip(Ws, fused):
        sep = gpu.mul_mat(W, xb, T.F32)
        assert torch.isfinite(f).all()
        assert torch.equal(_bits(f), _bits(sep)), (T.NAMES[W.type], W.rows)
