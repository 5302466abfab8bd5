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
from helpers import rel_err, elem_err

pytestmark = pytest.mark.gpu

# (weight type, m, k): attn_output / ffn_gate+up / ffn_down of Llama-3-8B Q4_K_M, the Q6_K ffn_down of its more-bits layers
SHAPES = [(T.Q4_K, 4096, 4096), (T.Q4_K, 14336, 4096), (T.Q4_K, 4096, 14336), (T.Q6_K, 4096, 14336)]
IDS = [f"{T.NAMES[t]}-{m}x{k}" for t, m, k in SHAPES]
# decode only: attn_k / attn_v, output.weight (Q6_K), and BASELINE config 3's Llama-3-8B Q8_0 shapes
DECODE_EXTRA = [(T.Q4_K, 1024, 4096), (T.Q6_K, 1024, 4096), (T.Q6_K, 128256, 4096), (T.Q8_0, 4096, 4096), (T.Q8_0, 1024, 4096),
                (T.Q8_0, 14336, 4096), (T.Q8_0, 4096, 14336)]
IDS_EXTRA = [f"{T.NAMES[t]}-{m}x{k}" for t, m, k in DECODE_EXTRA]
PREFILL = 512


def _weights(gpu, t, m, k, seed):
    from llamafile_amd import synth
    raw = synth.random_weights_torch(t, m, k, seed)  # uint8 [m, row_bytes] on the device
    return raw, gpu.upload_weights(t, raw, m, k)


def _bits(x):
    return x.contiguous().view(torch.int32)


@pytest.mark.parametrize("t,m,k", SHAPES + DECODE_EXTRA, ids=IDS + IDS_EXTRA)
def test_decode_gemv_full_size_vs_oracle(gpu, oracle, t, m, k):
    """Batch-1 GEMV at the model's real shapes against the CPU restatement (exact integer dots, f32 scales: 2e-6;
    Q8_0: bit for bit, the host's tinyBLAS_Q0 build)."""
    from llamafile_amd import synth, _hip
    raw, W = _weights(gpu, t, m, k, 11)
    x = synth.random_activations(1, k, 12)
    bt = T.VEC_DOT[t]
    B = synth.quantize_activations(bt, x)
    v = oracle.variant("zen4" if gpu.host_variant_flags() & _hip.FLAG_Q0_VREGS32 else "avx2")
    ok, G = oracle.sgemm(t, raw.cpu().numpy(), bt, B, m, 1, k, nth=8, v=v)
    assert ok == 1
    c_q = gpu.mul_mat(W, torch.from_numpy(B).cuda(), bt)
    c_f = gpu.mul_mat(W, torch.from_numpy(x).cuda().view(torch.uint8).view(1, k * 4), T.F32)
    torch.cuda.synchronize()
    if t == T.Q8_0:
        assert np.array_equal(c_q.cpu().numpy().view(np.uint32), G.view(np.uint32))
    else:
        assert rel_err(c_q.cpu().numpy(), G) <= 2e-6
        frac, worst = elem_err(c_q.cpu().numpy(), G, rtol=1e-5)
        assert frac == 0.0, (frac, worst)
    assert torch.equal(_bits(c_q), _bits(c_f))  # in-kernel quantisation == quantize_row_q8_K / q8_0


@pytest.mark.parametrize("t,m,k", [(T.Q4_K, 4096, 4096), (T.Q4_K, 4096, 14336), (T.Q4_K, 14336, 4096), (T.Q6_K, 4096, 14336), (T.Q5_K, 4096, 4096)],
                         ids=lambda v: str(v))
@pytest.mark.parametrize("dscale", ["synthetic", "real-model"])
def test_default_prefill_body_vs_oracle_deep_k(gpu, oracle, t, m, k, dscale):
    """The DEFAULT (benchmarked) prefill bodies against the ORACLE (not against the repo's own GEMV) at k = 4096 and 14336, on a
    bounded sample: 48 weight rows x 24 of 512 tokens.  Two weight magnitudes: the synthetic d in 2^-10..2^-6 and real-model-like d
    in 2^-15..2^-12 with small sub-block scales (f16 subnormal territory for d*sc).
      * Q4_K on grids the int8 body takes (4096-row matrices at 512 tokens; lfamd_mul_mat_is_exact): exact integer dots — 2e-6
        normwise AND no element beyond 1e-5 * (|G| + rms);
      * the scaled-operand f16 bodies (Q4_K on large grids: gemm_kr; Q5_K, Q6_K: gemm_lw): normwise <= 1e-3 (north star) AND no
        element beyond 1.5e-3 * (|G| + rms) (measured worst element 8.7-9.4e-4: profiles/r04_scaled_body_error.json).
    The measured numbers are printed for that record."""
    from llamafile_amd import _hip, synth
    n = 512
    rows = np.unique(np.concatenate([np.arange(0, m, m // 40), np.arange(4), np.arange(m - 4, m)]))[:48]
    raw = synth.random_weights(t, m, k, 4242)
    if dscale == "real-model":
        synth.rescale_blocks(t, raw, 2.0 ** -5)  # d, dmin x 2^-5: d ~ 2^-15..2^-11
    W = gpu.upload_weights(t, raw, m, k)
    x = synth.random_activations(n, k, 4343)
    C = gpu.mul_mat(W, torch.from_numpy(x).cuda().view(torch.uint8).view(n, k * 4), T.F32).cpu().numpy()
    cols = np.arange(0, n, 22)[:24]
    B = synth.quantize_activations(T.Q8_K, x[cols])
    ok, G = oracle.sgemm(t, np.ascontiguousarray(raw[rows]), T.Q8_K, B, len(rows), len(cols), k, nth=8)
    assert ok == 1
    got = C[np.ix_(cols, rows)]
    err = rel_err(got, G)
    exact = bool(_hip.lib().lfamd_mul_mat_is_exact(t, m, k, n, gpu.host_variant_flags() | (_hip.FLAG_PRECISE if W.exact_only else 0)))
    frac3, worst = elem_err(got, G, rtol=1e-3)
    frac, _ = elem_err(got, G, rtol=1e-5 if exact else 1.5e-3)
    print(f"DEFAULT_BODY_ERROR {T.NAMES[t]} m={m} k={k} d={dscale} body={'exact' if exact else 'scaled'}: normwise={err:.3e} "
          f"frac_over_1e-3={frac3:.4f} worst_elem={worst:.3e}")
    assert err <= (2e-6 if exact else 1e-3), err
    assert frac == 0.0, (frac, worst)


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
    as 128 x 64 tiles (the second matrix is split at a row-block boundary).  Against the two separate calls, which run
    128 x 128 tiles only (448 tiles each: no such tail): the rows that keep their tile shape are bit-identical; the Q4_K
    128 x 64 tail runs the K-split-waves body (gemm_ks.hip) since round 3, whose two wave groups sum their K halves
    separately — the same products in another f32 order (<= 4e-6 of the output scale; it was bit-identical on the
    loader-wave body, which LFAMD_GEMM_NO_KS=1 still selects)."""
    from llamafile_amd import synth
    t, m, k = T.Q4_K, 14336, 4096
    Ws = [gpu.upload_weights(t, synth.random_weights_torch(t, m, k, 31 + i), m, k) for i in range(2)]
    x = torch.from_numpy(synth.random_activations(PREFILL, k, 32)).cuda()
    xb = x.view(torch.uint8).view(PREFILL, k * 4)
    fused = gpu.mul_mat_multi(Ws, xb, T.F32, n=PREFILL)
    for W, f in zip(Ws, fused):
        sep = gpu.mul_mat(W, xb, T.F32)
        assert torch.isfinite(f).all()
        same = (_bits(f) == _bits(sep)).all(dim=0)  # per weight row
        assert float((f - sep).abs().max()) / float(sep.abs().max()) <= 4e-6
        # the tail = 128 tiles of 128 x 128 = 32 row blocks of the SECOND matrix
        assert int(same.sum()) >= m - (4096 if W is Ws[1] else 0), "rows outside the 128 x 64 tail changed"


@pytest.mark.parametrize("ta", [T.Q4_K, T.Q5_K], ids=lambda t: T.NAMES[t])
def test_qkv_two_types_one_gemm_launch(gpu, ta):
    """attn_q / attn_k (Q4_K or Q5_K) and attn_v (Q6_K) at 512 tokens through lfamd_mul_mat_multi_types: 192 tiles of
    128 x 128 in ONE launch whose work-groups run their own type's body.  Against the three separate calls (which run
    128 x 64 tiles): bit-identical where both run the loader-wave body (Q5_K, Q6_K); the separate Q4_K calls run the
    K-split-waves body (gemm_ks.hip, another f32 order of the same products: <= 4e-6 of the output scale) or, the 4096-row one, the
    int8 body (gemm_i8.hip: exact integer dots, so the scaled-operand launch is <= 1e-3 away)."""
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
        if W.type == T.Q4_K:
            # (a separate 4096 x 4096 call takes the int8 body since round 4 — exact integer dots — while the fused launch runs the
            # scaled-operand f16 body: one f16 rounding per operand apart)
            from llamafile_amd import _hip
            tol = 1e-3 if _hip.lib().lfamd_mul_mat_is_exact(W.type, W.rows, k, PREFILL, gpu.host_variant_flags()) else 4e-6
            assert float((f - sep).abs().max()) / float(sep.abs().max()) <= tol, (T.NAMES[W.type], W.rows)
        else:
            assert torch.equal(_bits(f), _bits(sep)), (T.NAMES[W.type], W.rows)


@pytest.mark.parametrize("case", [("gate", T.Q4_K, 14336, 4096), ("down", T.Q6_K, 4096, 14336)], ids=lambda c: c[0])
def test_mul_mat_id_full_size_mixtral(gpu, oracle, case):
    """BASELINE config 4 at its real sizes: GGML_OP_MUL_MAT_ID with 8 experts of 14336 x 4096 (Q4_K, ffn_gate_exps) and of
    4096 x 14336 (Q6_K, ffn_down_exps in a Q4_K_M file), 2 experts per token (shapes: tinyblas_cpu_mixmul.inc:65-72; call
    site ggml.c.patch:2004-2018).
      * decode (1 token): both chosen experts against the oracle;
      * 512 tokens: 24 sampled (token, thinker) rows against the oracle, and EVERY row against the decode GEMV of its expert on
        its activation row (the column-equals-GEMV property: the batch only regroups rows by expert);
      * an out-of-range expert id leaves its row untouched."""
    from llamafile_amd import synth, _hip
    _, t, rows, cols = case
    experts, thinkers, tasks, tokens = 8, 2, 1, PREFILL
    raws = [synth.random_weights_torch(t, rows, cols, 7000 + e) for e in range(experts)]
    Ws = [gpu.upload_weights(t, r, rows, cols) for r in raws]
    packed = torch.cat([W.data for W in Ws])
    x = synth.random_activations(tokens * tasks, cols, 71)
    xq = synth.quantize_activations(T.Q8_K, x)
    xd = torch.from_numpy(x).cuda()
    thought = xd.view(torch.uint8).view(tokens * tasks, cols * 4)
    rng = np.random.default_rng(72)
    plan = np.stack([rng.permutation(experts)[:thinkers] for _ in range(tokens)]).astype(np.int32)  # two DIFFERENT experts per token
    plan[17, 1] = experts + 1  # invalid id
    pd = torch.from_numpy(plan).cuda()

    def oracle_row(ex, tok):
        ok, G = oracle.sgemm(t, raws[ex].cpu().numpy(), T.Q8_K, xq[tok:tok + 1], rows, 1, cols, nth=8)
        assert ok == 1
        return G[0]

    # ---- decode: token 0 alone
    res1 = gpu.mul_mat_id(packed, t, rows, cols, experts, thought[:1].contiguous(), T.F32, tasks, 1, pd[:1].contiguous(), thinkers, prefill=-7.0)
    torch.cuda.synchronize()
    r1 = res1.cpu().numpy()
    for th in range(thinkers):
        assert rel_err(r1[0, th], oracle_row(int(plan[0, th]), 0)) <= 2e-6, th

    # ---- the 512-token batch (scaled operands: the default of every K-quant batch)
    res = gpu.mul_mat_id(packed, t, rows, cols, experts, thought, T.F32, tasks, tokens, pd, thinkers, prefill=-7.0)
    torch.cuda.synchronize()
    assert (res[17, 1] == -7.0).all()
    rn = res.cpu().numpy()
    scale = float(np.abs(rn[rn != -7.0]).max())
    picks = [(int(a), int(b)) for a, b in zip(rng.integers(0, tokens, 24), rng.integers(0, thinkers, 24)) if not (a == 17 and b == 1)]
    for tok, th in picks:
        g = oracle_row(int(plan[tok, th]), tok)
        assert float(np.abs(rn[tok, th] - g).max()) / scale <= 1e-3, (tok, th)
        assert rel_err(rn[tok, th], g) <= 1e-3, (tok, th)
    # every row == the decode GEMV of its expert on its activation row (exact integer dots), to the MFMA operand rounding
    worst = 0.0
    for ex in range(experts):
        sel = [(tok, th) for tok in range(tokens) for th in range(thinkers) if plan[tok, th] == ex]
        toks = torch.tensor([tok for tok, _ in sel], device="cuda")
        got = torch.stack([res[tok, th] for tok, th in sel])
        for c0 in range(0, len(sel), 8):  # the GEMV takes up to 8 columns per call
            idx = toks[c0:c0 + 8]
            g = gpu.mul_mat(Ws[ex], thought[idx].contiguous(), T.F32, n=len(idx))
            worst = max(worst, float((got[c0:c0 + 8] - g).abs().max()) / scale)
    assert worst <= 1e-3, worst
