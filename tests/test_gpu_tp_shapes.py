"""BASELINE config 5 (Llama-3-70B Q4_K_M, tensor parallel over 8 GPUs) and the TP shards of both models through the
device C ABI: decode (n = 1) against the oracle, n = 512 through a property the domain offers — a GEMM column equals
the GEMV of that column (the reference's CPU path computes each output independently, iqk_mul_mat.inc:110-127).
Shard rule: SURVEY.md §8e (rows for attn_q/k/v, ffn_gate/up, output; k for attn_output, ffn_down; k / N a multiple of
the 256-element super-block).  Reference counterpart: the row-split of ggml-cuda.cu.patch:17123-17450, 17853-18153."""
import numpy as np
import pytest
import torch

from llamafile_amd import ggml_types as T, llama_shapes as LS, synth
from helpers import rel_err, elem_err

pytestmark = pytest.mark.gpu

# (type, m, k): full 70B shapes, then the shards of 8B and 70B at N = 2 and N = 8 (incl. odd super-block counts:
# k = 1792 = 7, k = 3584 = 14 super-blocks; m = 128 rows = one kv head)
FULL_70B = [(T.Q4_K, 8192, 8192), (T.Q6_K, 1024, 8192), (T.Q4_K, 28672, 8192), (T.Q4_K, 8192, 28672), (T.Q6_K, 8192, 28672)]
SHARDS = [(T.Q4_K, 512, 4096), (T.Q4_K, 128, 4096), (T.Q6_K, 128, 4096), (T.Q4_K, 1792, 4096), (T.Q4_K, 4096, 512),
          (T.Q4_K, 4096, 1792), (T.Q6_K, 4096, 1792), (T.Q6_K, 16032, 4096), (T.Q4_K, 1024, 8192), (T.Q4_K, 3584, 8192),
          (T.Q4_K, 8192, 1024), (T.Q4_K, 8192, 3584), (T.Q6_K, 8192, 3584), (T.Q4_K, 2048, 4096), (T.Q4_K, 7168, 4096),
          (T.Q4_K, 4096, 2048), (T.Q4_K, 4096, 7168), (T.Q4_K, 14336, 8192), (T.Q4_K, 8192, 14336)]


def sub_rows(raw, rows):
    return np.ascontiguousarray(raw[rows])


@pytest.mark.parametrize("t,m,k", FULL_70B + SHARDS, ids=[f"{T.NAMES[t]}-{m}x{k}" for t, m, k in FULL_70B + SHARDS])
def test_decode_vs_oracle(gpu, oracle, t, m, k):
    raw = synth.random_weights_torch(t, m, k, 1000 + m % 977 + k)
    W = gpu.upload_weights(t, raw, m, k)
    x = synth.random_activations(1, k, 5)
    out = gpu.mul_mat(W, torch.from_numpy(x).cuda().view(torch.uint8).view(1, k * 4), T.F32, n=1).cpu().numpy()
    # the oracle on a bounded sample of rows (all of a small matrix; every 37th row plus both ends of a big one)
    rows = np.arange(m) if m <= 2048 else np.unique(np.concatenate([np.arange(0, m, 37), np.arange(64), np.arange(m - 64, m)]))
    B = synth.quantize_activations(T.Q8_K, x)
    ok, G = oracle.sgemm(t, sub_rows(raw.cpu().numpy(), rows), T.Q8_K, B, len(rows), 1, k, nth=4)
    assert ok == 1
    assert rel_err(out[:, rows], G) <= 2e-6
    frac, worst = elem_err(out[:, rows], G, rtol=1e-5)
    assert frac == 0.0, (frac, worst)


@pytest.mark.parametrize("t,m,k", [(T.Q4_K, 8192, 8192), (T.Q4_K, 28672, 8192), (T.Q4_K, 8192, 28672), (T.Q4_K, 128, 4096),
                                   (T.Q4_K, 4096, 512), (T.Q4_K, 4096, 1792), (T.Q6_K, 4096, 1792), (T.Q4_K, 8192, 3584),
                                   (T.Q4_K, 1024, 8192), (T.Q4_K, 3584, 8192)], ids=lambda v: str(v))
def test_prefill_columns_equal_decode(gpu, t, m, k):
    """n = 512 on the default (scaled-operand MFMA) body: sampled columns against the exact-integer GEMV of the same
    column, within the stated 1e-3 (normwise) and with the element-wise statistic."""
    n = 512
    W = gpu.upload_weights(t, synth.random_weights_torch(t, m, k, 2000 + m % 977 + k), m, k)
    x = torch.from_numpy(synth.random_activations(n, k, 6)).cuda()
    C = gpu.mul_mat(W, x.view(torch.uint8), T.F32, n=n)
    cols = [0, 1, 129, 255, 256, 511]
    worst = 0.0
    for c in cols:
        g = gpu.mul_mat(W, x[c:c + 1].contiguous().view(torch.uint8), T.F32, n=1)
        worst = max(worst, rel_err(C[c:c + 1].cpu().numpy(), g.cpu().numpy()))
        frac, _ = elem_err(C[c:c + 1].cpu().numpy(), g.cpu().numpy(), rtol=2e-3)
        assert frac <= 1e-3, (c, frac)
    assert worst <= 1e-3, worst


def test_llama3_70b_layer_table():
    layers = LS.llama3_70b_q4_k_m()
    assert len(layers) == 81
    assert {(s.m, s.k) for s in layers[0]} == {(8192, 8192), (1024, 8192), (28672, 8192), (8192, 28672)}
    for world in (2, 4, 8):
        for s in layers[0] + layers[-1]:
            if s.shard in ("rows", "vocab"):
                assert s.m % world == 0
            else:
                assert s.k % (world * 256) == 0
