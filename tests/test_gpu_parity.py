"""GPU parity: HIP kernels (through the C ABI) vs the CPU oracle on identical quantised bytes."""
import numpy as np
import pytest
import torch

from llamafile_amd import ggml_types as T
from helpers import make_case, rel_err

pytestmark = pytest.mark.gpu

# integer parts are exact on both sides; only f32 scale products / summation order differ
TOL = {T.Q4_1: 1e-5, T.Q5_1: 1e-5}
DEFAULT_TOL = 2e-6
# north-star tolerance: logits within 1e-3 relative; Q6_K's MFMA operand rounds sc*(q-32) > 2048
GEMM_TOL = {T.Q4_K: 2e-6, T.Q5_K: 2e-6, T.Q6_K: 1e-3}
# Q4_K / Q5_K on the loader-wave body without LFAMD_FLAG_PRECISE: scaled operands, one f16 rounding of d*sc*q and of
# d8*code each (gemm_lw.hip FAST); measured 2.5e-4 .. 5e-4 at k = 256 .. 768
SCALED_TOL = 1e-3


def gemm_tol(t, body_flags, n):
    from llamafile_amd import _hip
    if n <= 8:
        return DEFAULT_TOL
    # default route and LFAMD_FLAG_GEMM_WIDE: the loader-wave body on scaled operands at every grid size
    scaled = (t in (T.Q4_K, T.Q5_K, T.Q6_K) and not (body_flags & (_hip.FLAG_GEMM_PLAIN | _hip.FLAG_PRECISE | _hip.FLAG_GEMM_NARROW)))
    return SCALED_TOL if scaled else GEMM_TOL.get(t, DEFAULT_TOL)


def run_gpu(gpu, t, A, B, bt, m, n, k, flags=None):
    W = gpu.upload_weights(t, A, m, k)
    Bd = torch.from_numpy(B).cuda()
    C = gpu.mul_mat(W, Bd, bt, flags=flags)
    torch.cuda.synchronize()
    return C.cpu().numpy()


@pytest.mark.parametrize("t", T.QUANT_WEIGHT_TYPES, ids=lambda t: T.NAMES[t])
@pytest.mark.parametrize("shape", [(64, 1, 1024), (67, 5, 512), (128, 8, 768), (33, 3, 256)], ids=str)
def test_small_n_vs_oracle(gpu, oracle, t, shape):
    m, n, k = shape
    A, B, bt = make_case(t, m, n, k, seed=100 + t)
    ok, G = oracle.sgemm(t, A, bt, B, m, n, k, nth=3)
    assert ok == 1 and not np.isnan(G).any()
    C = run_gpu(gpu, t, A, B, bt, m, n, k)
    assert not np.isnan(C).any()
    if t == T.Q8_0:
        return  # bit-exact test below
    assert rel_err(C, G) <= TOL.get(t, DEFAULT_TOL), (T.NAMES[t], shape, rel_err(C, G))


@pytest.mark.parametrize("variant", ["zen4", "avx2"])
@pytest.mark.parametrize("precise", [0, 1])
@pytest.mark.parametrize("shape", [(64, 1, 1024), (37, 5, 4096), (128, 8, 768), (9, 2, 32), (5, 7, 96), (100, 24, 256),
                                   (70, 45, 224), (33, 100, 544), (257, 33, 512), (31, 9, 32)], ids=str)
def test_q8_0_bit_exact(gpu, oracle, variant, precise, shape):
    """Q8_0 vecdot bit-exact (north star): every output equals the restated tinyBLAS_Q0_AVX2 bit for
    bit, for both reference builds (32 / 16 vector registers) and --precise."""
    from llamafile_amd import _hip
    m, n, k = shape
    A, B, bt = make_case(T.Q8_0, m, n, k, seed=7)
    v = oracle.variant(variant, precise=precise)
    ok, G = oracle.sgemm(T.Q8_0, A, bt, B, m, n, k, v=v, nth=2)
    assert ok == 1
    # (batches default to the library GEMM or the MFMA body — test_q8_0_batches_default; the bit-exact kernel is asked for explicitly)
    flags = (_hip.FLAG_Q0_VREGS32 if variant == "zen4" else 0) | (_hip.FLAG_PRECISE if precise else 0) | _hip.FLAG_Q80_EXACT
    C = run_gpu(gpu, T.Q8_0, A, B, bt, m, n, k, flags=flags)
    # Reference quirk: the 16-vector-register --precise build switches on MIN(n - n0, 1)
    # (tinyblas_cpu.h:904) and so runs gemm<2,1>/<1,1> with xtiles = 1: for n > 1 it writes column 0
    # only and leaves the rest of C untouched.  The oracle restates that faithfully (NaN prefill
    # survives); the GPU computes every column.  Compare where the reference writes.
    written = oracle.q0_precise_map(m, n, v) != 0xFF
    if not (variant == "avx2" and precise):
        assert written.all()
    assert np.array_equal(C.view(np.uint32)[written], G.view(np.uint32)[written]), np.abs(C - G)[written].max()
    assert not np.isnan(C).any()


@pytest.mark.parametrize("shape", [(128, 64, 512), (96, 100, 1024), (33, 9, 256), (256, 512, 2048), (64, 130, 768), (300, 40, 4096), (64, 20, 128),
                                   (40, 33, 384), (1100, 200, 640), (72, 12, 160)], ids=str)
@pytest.mark.parametrize("f32in", [False, True], ids=["q80", "f32"])
def test_q8_0_batches_default(gpu, oracle, shape, f32in):
    """Q8_0, n > 8, default flags: the module's f16 MFMA body on the resident P80 image (gemm_lf.hip): f16(d * q) x f16(d8 * code),
    <= 1e-3 normwise like the scaled K-quant batches and no element beyond 3e-3 (|G| + rms); rows that are not whole 128-weight
    quads run the bit-exact kernel.  A process that opted into the vendor GEMM (LFAMD_USE_BLASLT=1: the subprocess test below) runs
    hipBLASLt on a second, f16 image — the same arithmetic.  LFAMD_FLAG_Q80_EXACT gives the bit-exact kernel (test_q8_0_bit_exact)."""
    from llamafile_amd import synth
    from helpers import q80_batch_tol
    m, n, k = shape
    A = synth.random_weights(T.Q8_0, m, k, 31)
    x = synth.random_activations(n, k, 32)
    B = synth.quantize_activations(T.Q8_0, x)
    ok, G = oracle.sgemm(T.Q8_0, A, T.Q8_0, B, m, n, k, nth=4)
    assert ok == 1
    W = gpu.upload_weights(T.Q8_0, A, m, k)
    Bd = torch.from_numpy(x).cuda().view(torch.uint8).view(n, k * 4) if f32in else torch.from_numpy(B).cuda()
    C = gpu.mul_mat(W, Bd, T.F32 if f32in else T.Q8_0).cpu().numpy()
    tol = q80_batch_tol(m, k, n)
    assert rel_err(C, G) <= tol, rel_err(C, G)
    from helpers import elem_err
    frac, worst = elem_err(C, G, rtol=1e-5 if tol < 1e-5 else 3e-3)
    assert frac == 0.0, (frac, worst)


def test_q8_0_batches_with_the_vendor_library(gpu):
    """The same cases in a process that opted into hipBLASLt (LFAMD_USE_BLASLT=1): the library GEMM on the f16(d * q) image, <= 1e-3.
    The default (this process) is the module's own MFMA body on the resident P80 image (the same arithmetic)."""
    import os
    import subprocess
    import sys
    if os.environ.get("LFAMD_USE_BLASLT"):
        pytest.skip("already the opted-in process")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", __file__, "-k", "test_q8_0_batches_default or test_float_types_mfma_gemm",
                        "-p", "no:cacheprovider"], env={**os.environ, "LFAMD_USE_BLASLT": "1"}, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("t", [T.Q4_K, T.Q5_K, T.Q6_K], ids=lambda t: T.NAMES[t])
@pytest.mark.parametrize("shape", [(128, 64, 512), (96, 100, 1024), (33, 9, 256), (256, 512, 2048), (64, 130, 768)],
                         ids=str)
@pytest.mark.parametrize("body", ["narrow", "wide", "wide_plain", "wide_exact"])
def test_mfma_gemm_vs_oracle(gpu, oracle, t, shape, body):
    """Both MFMA bodies (128x64 split-K, gemm_mfma.hip; 128x128, gemm_wide.hip) on every shape, ragged m / n
    included; the default picks by grid size, which small test shapes would never send to the wide body."""
    from llamafile_amd import _hip
    m, n, k = shape
    A, B, bt = make_case(t, m, n, k, seed=300 + t)
    ok, G = oracle.sgemm(t, A, bt, B, m, n, k, nth=4)
    assert ok == 1
    # "wide": Q4_K / Q5_K run the loader-wave body (gemm_lw.hip) with scaled operands when K is not split;
    # "wide_exact": the same body on exact integer codes; "wide_plain": the same tile without loader waves
    flag = {"narrow": _hip.FLAG_GEMM_NARROW, "wide": _hip.FLAG_GEMM_WIDE, "wide_plain": _hip.FLAG_GEMM_WIDE | _hip.FLAG_GEMM_PLAIN,
            "wide_exact": _hip.FLAG_GEMM_WIDE | _hip.FLAG_PRECISE}[body]
    C = run_gpu(gpu, t, A, B, bt, m, n, k, flags=gpu.host_variant_flags() | flag)
    assert not np.isnan(C).any()
    assert rel_err(C, G) <= gemm_tol(t, flag, n), (T.NAMES[t], shape, body, rel_err(C, G))


@pytest.mark.parametrize("t", [T.Q4_K, T.Q5_K, T.Q6_K], ids=lambda t: T.NAMES[t])
@pytest.mark.parametrize("f32in", [False, True], ids=["q8k", "f32"])
def test_loader_wave_gemm_full_grid(gpu, oracle, t, f32in):
    """The default route of a grid that fills the chip (>= 192 tiles, K not split): loader-wave body, scaled operands by
    default and exact codes with LFAMD_FLAG_PRECISE; ragged rows and tokens, 6 super-blocks."""
    from llamafile_amd import _hip, synth
    m, n, k = 1000, 3100, 1536
    A = synth.random_weights(t, m, k, 4100 + t)
    x = synth.random_activations(n, k, 4101)
    B = synth.quantize_activations(T.Q8_K, x)
    ok, G = oracle.sgemm(t, A, T.Q8_K, B, m, n, k, nth=8)
    assert ok == 1
    W = gpu.upload_weights(t, A, m, k)
    Bd = torch.from_numpy(x).cuda().view(torch.uint8).view(n, k * 4) if f32in else torch.from_numpy(B).cuda()
    for flags, tol in ((0, SCALED_TOL), (_hip.FLAG_PRECISE, GEMM_TOL[t])):
        C = gpu.mul_mat(W, Bd, T.F32 if f32in else T.Q8_K, flags=gpu.host_variant_flags() | flags)
        torch.cuda.synchronize()
        err = rel_err(C.cpu().numpy(), G)
        assert err <= tol, (T.NAMES[t], flags, err)


@pytest.mark.parametrize("t", [T.Q4_K, T.Q5_K, T.Q6_K], ids=lambda t: T.NAMES[t])
@pytest.mark.parametrize("shape", [(4096, 9, 4096), (4096, 32, 4096), (1000, 100, 2048), (2048, 128, 1024), (300, 70, 5 * 256),
                                   (8192, 64, 2048)], ids=str)
def test_few_token_batches_k_split(gpu, oracle, t, shape):
    """Batches of 9 .. 128 tokens on one matrix: the scaled-operand body cuts K into up to 8 parts across work-groups
    (partial tiles in the workspace, summed in a fixed order by a second kernel) — against the oracle, bit-identical on a
    rerun, ragged rows / tokens, odd super-block counts, and shapes where the split does not apply."""
    from llamafile_amd import synth
    m, n, k = shape
    A = synth.random_weights(t, m, k, 7100 + t)
    x = synth.random_activations(n, k, 7101)
    B = synth.quantize_activations(T.Q8_K, x)
    ok, G = oracle.sgemm(t, A, T.Q8_K, B, m, n, k, nth=8)
    assert ok == 1
    W = gpu.upload_weights(t, A, m, k)
    Bd = torch.from_numpy(x).cuda().view(torch.uint8).view(n, k * 4)
    C1 = gpu.mul_mat(W, Bd, T.F32)
    C2 = gpu.mul_mat(W, Bd, T.F32)
    torch.cuda.synchronize()
    assert torch.equal(C1.view(torch.int32), C2.view(torch.int32))
    assert rel_err(C1.cpu().numpy(), G) <= SCALED_TOL, (T.NAMES[t], shape, rel_err(C1.cpu().numpy(), G))


@pytest.mark.parametrize("t", [T.Q4_K, T.Q6_K], ids=lambda t: T.NAMES[t])
@pytest.mark.parametrize("f32in", [False, True], ids=["q8k", "f32"])
def test_scaled_gemm_activation_range(gpu, oracle, t, f32in):
    """The scaled-operand body normalises every token by a power of two before the f16 staging (prep_scaled_kernel) and
    undoes it in the store: tokens of magnitude 1e-7 .. 3e5 (far outside f16's 6e-5 .. 65504), an all-zero token and
    ordinary ones in one batch all meet the north-star tolerance per token."""
    from llamafile_amd import synth
    m, n, k = 1000, 3100, 1024
    A = synth.random_weights(t, m, k, 6100 + t)
    x = synth.random_activations(n, k, 6101)
    mags = np.ones(n, dtype=np.float32)
    mags[0::5] = 1e-7
    mags[1::5] = 3e5
    mags[2::5] = 2e-3
    x = x * mags[:, None]
    x[7] = 0.0
    x[8, 100:] *= 1e-6  # one token whose super-blocks differ by six orders of magnitude
    B = synth.quantize_activations(T.Q8_K, x)
    ok, G = oracle.sgemm(t, A, T.Q8_K, B, m, n, k, nth=8)
    assert ok == 1
    W = gpu.upload_weights(t, A, m, k)
    Bd = torch.from_numpy(x).cuda().view(torch.uint8).view(n, k * 4) if f32in else torch.from_numpy(B).cuda()
    C = gpu.mul_mat(W, Bd, T.F32 if f32in else T.Q8_K).cpu().numpy()
    assert np.isfinite(C).all()
    assert (C[7] == 0).all()
    for j in list(range(0, 40)) + [n - 2, n - 1]:  # per token: each against its own scale
        assert rel_err(C[j], G[j]) <= SCALED_TOL, (j, mags[j], rel_err(C[j], G[j]))


@pytest.mark.parametrize("t", [T.Q4_K, T.Q5_K], ids=lambda t: T.NAMES[t])
def test_out_of_range_scales_run_exact(gpu, oracle, t):
    """lfamd_scaled_gemm_ok: block scales beyond the scaled-operand body's f16 range (|d| * 63 >= 64) are detected at
    upload and the matrix then always takes the exact integer-code path; in-range matrices report 1."""
    from llamafile_amd import synth
    m, n, k = 1000, 3100, 1024
    A = synth.random_weights(t, m, k, 5100 + t).copy()
    blk = T.row_size(t, 256)
    hdr = A.reshape(m, k // 256, blk)[:, :, :2].view(np.float16)  # d of every super-block
    assert gpu.upload_weights(t, A, m, k).exact_only is False
    hdr[17, 2, 0] = np.float16(1.5)
    x = synth.random_activations(n, k, 5101)
    B = synth.quantize_activations(T.Q8_K, x)
    ok, G = oracle.sgemm(t, A, T.Q8_K, B, m, n, k, nth=8)
    assert ok == 1
    W = gpu.upload_weights(t, A, m, k)
    assert W.exact_only is True
    C = gpu.mul_mat(W, torch.from_numpy(B).cuda(), T.Q8_K)
    torch.cuda.synchronize()
    assert rel_err(C.cpu().numpy(), G) <= GEMM_TOL[t]


@pytest.mark.parametrize("t", [T.Q4_K, T.Q5_K, T.Q6_K], ids=lambda t: T.NAMES[t])
def test_mfma_bodies_agree_and_repeat(gpu, t):
    """Same inputs -> the wide body gives identical bits on every run (no atomics in the default K split), and
    the two bodies agree to the f32 summation-order tolerance.  Odd super-block count, ragged rows and tokens."""
    from llamafile_amd import _hip
    m, n, k = 300, 200, 256 * 5
    A, B, bt = make_case(t, m, n, k, seed=77)
    base = gpu.host_variant_flags()
    w1 = run_gpu(gpu, t, A, B, bt, m, n, k, flags=base | _hip.FLAG_GEMM_WIDE)
    w2 = run_gpu(gpu, t, A, B, bt, m, n, k, flags=base | _hip.FLAG_GEMM_WIDE)
    nr = run_gpu(gpu, t, A, B, bt, m, n, k, flags=base | _hip.FLAG_GEMM_NARROW)
    assert np.array_equal(w1.view(np.uint32), w2.view(np.uint32))
    assert rel_err(w1, nr) <= gemm_tol(t, _hip.FLAG_GEMM_WIDE, n)
    # exact integer codes on both tiles: only the f32 summation order differs
    we = run_gpu(gpu, t, A, B, bt, m, n, k, flags=base | _hip.FLAG_GEMM_WIDE | _hip.FLAG_PRECISE)
    assert rel_err(we, nr) <= 2e-6


@pytest.mark.parametrize("t", [T.Q2_K, T.Q3_K, T.IQ4_XS], ids=lambda t: T.NAMES[t])
@pytest.mark.parametrize("shape", [(128, 64, 512), (96, 100, 1024), (33, 9, 256), (200, 300, 1280), (64, 130, 768)], ids=str)
@pytest.mark.parametrize("f32in", [False, True], ids=["q8k", "f32"])
def test_canonical_image_gemm_vs_oracle(gpu, oracle, t, shape, f32in):
    """Q2_K / Q3_K batches: RAW weights are canonicalised per call (wprep16 -> PCK image) and run on the 128x128 MFMA
    body; integer parts exact (sc*q <= 128), Q2_K mins through one MFMA on the 16 bsums."""
    from llamafile_amd import synth
    m, n, k = shape
    A = synth.random_weights(t, m, k, 500 + t)
    x = synth.random_activations(n, k, 501)
    B = synth.quantize_activations(T.Q8_K, x)
    ok, G = oracle.sgemm(t, A, T.Q8_K, B, m, n, k, nth=2)
    assert ok == 1
    W = gpu.upload_weights(t, A, m, k)
    if f32in:
        C = gpu.mul_mat(W, torch.from_numpy(x).cuda().view(torch.uint8).view(n, k * 4), T.F32)
    else:
        C = gpu.mul_mat(W, torch.from_numpy(B).cuda(), T.Q8_K)
    torch.cuda.synchronize()
    # IQ4_XS: |sc * kvalue| reaches 4064, products above 2048 round to even in f16 (like Q6_K): north-star tolerance
    assert rel_err(C.cpu().numpy(), G) <= (1e-3 if t == T.IQ4_XS else DEFAULT_TOL)


@pytest.mark.parametrize("t", [T.Q4_0, T.Q5_K, T.IQ4_XS, T.Q2_K], ids=lambda t: T.NAMES[t])
def test_generic_large_n(gpu, oracle, t):
    m, n, k = 64, 40, 512
    A, B, bt = make_case(t, m, n, k, seed=500 + t)
    ok, G = oracle.sgemm(t, A, bt, B, m, n, k)
    assert ok == 1
    C = run_gpu(gpu, t, A, B, bt, m, n, k)
    assert rel_err(C, G) <= (1e-3 if t in (T.IQ4_XS, T.Q5_K) else DEFAULT_TOL)  # Q5_K batch: scaled f16 operands


@pytest.mark.parametrize("vdt", [T.Q8_0, T.Q8_1, T.Q8_K], ids=lambda t: T.NAMES[t])
def test_device_quantiser_bit_exact(gpu, oracle, vdt):
    """lfamd_quantize_rows == the scalar reference quantisers (restated in the oracle), byte for byte,
    including an all-zero block and a tie between +max and -max (first occurrence wins in q8_K)."""
    from llamafile_amd import synth
    x = synth.random_activations(7, 1024, 17)
    x[2, 256:512] = 0.0
    x[3, 5] = 0.75
    x[3, 9] = -0.75
    x[3, :256] = np.clip(x[3, :256], -0.75, 0.75)
    got = gpu.quantize_rows(vdt, torch.from_numpy(x).cuda()).cpu().numpy()
    want = oracle.quantize(vdt, x)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("t", [T.Q4_K, T.Q6_K, T.Q8_0, T.Q4_0, T.Q5_1, T.Q3_K], ids=lambda t: T.NAMES[t])
@pytest.mark.parametrize("n", [1, 3, 8, 70])
def test_f32_activations_equal_prequantised(gpu, t, n):
    """GGML_OP_MUL_MAT boundary (f32 src1): quantisation fused into the kernels must give bit-identical
    results to quantising first (the llamafile_sgemm boundary)."""
    from llamafile_amd import synth
    m, k = 96, 1536
    A = synth.random_weights(t, m, k, 21)
    x = synth.random_activations(n, k, 22)
    x[0, :256] = 0.0
    bt = T.VEC_DOT[t]
    W = gpu.upload_weights(t, A, m, k)
    xd = torch.from_numpy(x).cuda()
    c_f32 = gpu.mul_mat(W, xd.view(torch.uint8), T.F32).cpu().numpy()
    c_q = gpu.mul_mat(W, gpu.quantize_rows(bt, xd), bt).cpu().numpy()
    assert not np.isnan(c_f32).any()
    assert np.array_equal(c_f32.view(np.uint32), c_q.view(np.uint32))


@pytest.mark.parametrize("t", [T.Q4_K, T.Q5_K, T.Q6_K, T.Q8_0, T.Q4_0], ids=lambda t: T.NAMES[t])
@pytest.mark.parametrize("n", [1, 4, 20])
def test_mul_mat_multi_equals_separate(gpu, t, n):
    """Sibling mat-muls fused into one launch (attn_q/k/v, ffn_gate/up) give bit-identical results to
    separate lfamd_mul_mat calls — including odd row counts and a matrix smaller than a tile."""
    from llamafile_amd import synth
    k = 1024
    ms = [96, 40, 7, 130]
    Ws = [gpu.upload_weights(t, synth.random_weights(t, m, k, 60 + i), m, k) for i, m in enumerate(ms)]
    x = torch.from_numpy(synth.random_activations(n, k, 70)).cuda()
    fused = gpu.mul_mat_multi(Ws, x.view(torch.uint8), T.F32, n=n)
    for W, f in zip(Ws, fused):
        sep = gpu.mul_mat(W, x.view(torch.uint8), T.F32, n=n)
        if n == 1 or n > 8:
            assert np.array_equal(f.cpu().numpy().view(np.uint32), sep.cpu().numpy().view(np.uint32))
        else:  # a few tokens: a lone matrix may take the small-batch MFMA kernel where the group stays on the fused GEMV
            assert rel_err(f.cpu().numpy(), sep.cpu().numpy()) <= 2e-6  # (exact integer dots, f32 scales either way)


@pytest.mark.parametrize("n", [6, 8, 16])
def test_mul_mat_multi_small_batch_shares_the_staging(gpu, n):
    """Several tokens on sibling matrices with several row tiles per CU (ffn_gate + ffn_up): the activations are staged once and
    every matrix runs the small-batch MFMA kernel on them — the bits of separate lfamd_mul_mat calls."""
    from llamafile_amd import synth
    k, ms = 512, [9000, 8300]
    Ws = [gpu.upload_weights(T.Q4_K, synth.random_weights_torch(T.Q4_K, m, k, 80 + i).cpu().numpy(), m, k) for i, m in enumerate(ms)]
    x = torch.from_numpy(synth.random_activations(n, k, 81)).cuda()
    fused = gpu.mul_mat_multi(Ws, x.view(torch.uint8), T.F32, n=n)
    for W, f in zip(Ws, fused):
        sep = gpu.mul_mat(W, x.view(torch.uint8), T.F32, n=n)
        assert torch.equal(f.view(torch.int32), sep.view(torch.int32))


@pytest.mark.parametrize("ta", [T.Q4_K, T.Q5_K], ids=lambda t: T.NAMES[t])
@pytest.mark.parametrize("k", [1024, 4096, 5120])
@pytest.mark.parametrize("f32in", [True, False], ids=["f32", "q8k"])
def test_mul_mat_multi_two_types_one_launch(gpu, ta, k, f32in):
    """lfamd_mul_mat_multi_types at decode: attn_q/k (Q4_K or Q5_K) and attn_v (Q6_K) on one activation row run as one
    launch of the two-type GEMV; the same integer block dots as separate calls, in any node order, ragged row counts, both
    chunk depths (k <= 4096 / beyond).  The f32 sums over a row's super-blocks are ordered by the launch's wave layout (8
    waves x 2 blocks for a launch of at most one half-tile per CU, 16 waves otherwise), which a fused and a separate launch
    may pick differently: equal to f32 rounding (<= 1e-6 of the largest output), bit-identical when the layouts agree.
    Other mixes and batches fall back to per-type calls with the same results."""
    from llamafile_amd import synth
    specs = [(ta, 200), (ta, 40), (T.Q6_K, 72), (ta, 7), (T.Q6_K, 130)]
    Ws = [gpu.upload_weights(t, synth.random_weights(t, m, k, 260 + i), m, k) for i, (t, m) in enumerate(specs)]
    x = torch.from_numpy(synth.random_activations(3, k, 270)).cuda()
    for n in (1, 3):
        B = x[:n].contiguous()
        if f32in:
            Bq, bt = B.view(torch.uint8), T.F32
        else:
            Bq, bt = gpu.quantize_rows(T.Q8_K, B), T.Q8_K
        fused = gpu.mul_mat_multi(Ws, Bq, bt, n=n)
        for W, f in zip(Ws, fused):
            sep = gpu.mul_mat(W, Bq, bt, n=n)
            assert rel_err(f.cpu().numpy(), sep.cpu().numpy()) <= 1e-6, (n, T.NAMES[W.type], W.rows)
    # a third type in the mix: per-type fallback
    Wx = Ws[:3] + [gpu.upload_weights(T.Q8_0, synth.random_weights(T.Q8_0, 24, k, 299), 24, k)]
    fused = gpu.mul_mat_multi(Wx, x[:1].contiguous().view(torch.uint8), T.F32, n=1)
    for W, f in zip(Wx, fused):
        sep = gpu.mul_mat(W, x[:1].contiguous().view(torch.uint8), T.F32, n=1)
        assert np.array_equal(f.cpu().numpy().view(np.uint32), sep.cpu().numpy().view(np.uint32))


@pytest.mark.parametrize("ta", [T.Q4_K, T.Q5_K], ids=lambda t: T.NAMES[t])
@pytest.mark.parametrize("f32in", [True, False], ids=["f32", "q8k"])
def test_mul_mat_multi_two_types_batch_shares_the_prep(gpu, oracle, ta, f32in):
    """lfamd_mul_mat_multi_types for a batch: attn_q/k (Q4_K or Q5_K) and attn_v (Q6_K) read ONE staged copy of the
    activations (scaled operands), each type run is one launch of the loader-wave body; every output against the oracle."""
    from llamafile_amd import synth
    k, n = 1024, 150
    specs = [(ta, 200), (ta, 40), (T.Q6_K, 72)]
    raws = [synth.random_weights(t, m, k, 360 + i) for i, (t, m) in enumerate(specs)]
    Ws = [gpu.upload_weights(t, r, m, k) for (t, m), r in zip(specs, raws)]
    x = synth.random_activations(n, k, 370)
    B = synth.quantize_activations(T.Q8_K, x)
    Bd = torch.from_numpy(x).cuda().view(torch.uint8).view(n, k * 4) if f32in else torch.from_numpy(B).cuda()
    outs = gpu.mul_mat_multi(Ws, Bd, T.F32 if f32in else T.Q8_K, n=n)
    torch.cuda.synchronize()
    for (t, m), r, o in zip(specs, raws, outs):
        ok, G = oracle.sgemm(t, r, T.Q8_K, B, m, n, k, nth=4)
        assert ok == 1
        assert rel_err(o.cpu().numpy(), G) <= SCALED_TOL, (T.NAMES[t], m, rel_err(o.cpu().numpy(), G))


@pytest.mark.parametrize("t", [T.Q4_K, T.Q5_K, T.Q6_K], ids=lambda t: T.NAMES[t])
def test_mul_mat_multi_gemm_fused_launch(gpu, t):
    """Batches: sibling mat-muls share one activation prep and ONE launch of the 128 x 128 MFMA body over their
    concatenated row blocks; bit-identical to separate calls of the same body (ragged rows, a 7-row matrix)."""
    from llamafile_amd import synth, _hip
    k, n = 768, 150
    ms = [200, 128, 7, 130]
    Ws = [gpu.upload_weights(t, synth.random_weights(t, m, k, 160 + i), m, k) for i, m in enumerate(ms)]
    x = torch.from_numpy(synth.random_activations(n, k, 170)).cuda()
    flags = gpu.host_variant_flags() | _hip.FLAG_GEMM_WIDE
    fused = gpu.mul_mat_multi(Ws, x.view(torch.uint8), T.F32, n=n, flags=flags)
    for W, f in zip(Ws, fused):
        sep = gpu.mul_mat(W, x.view(torch.uint8), T.F32, n=n, flags=flags)
        assert np.array_equal(f.cpu().numpy().view(np.uint32), sep.cpu().numpy().view(np.uint32))


@pytest.mark.parametrize("ta,tb", [(T.F32, T.F32), (T.F16, T.F16), (T.F16, T.F32), (T.BF16, T.BF16), (T.BF16, T.F32)],
                         ids=lambda t: T.NAMES[t])
@pytest.mark.parametrize("shape", [(64, 1, 1000), (33, 2, 513), (40, 17, 256)], ids=str)
def test_float_types_vs_oracle(gpu, oracle, ta, tb, shape):
    """tinyBLAS<> float path (tinyblas_cpu.h:419-613; TinyLLama-F16 plumbing config): f32 accumulation, so
    the GPU result is compared with the reference's own criterion — a double-accumulator GEMM."""
    from llamafile_amd import synth
    m, n, k = shape
    if tb == T.F32 and ta != T.F32 and n > 2:
        pytest.skip("the reference declines F16/BF16 x F32 for n > 2 (WANT_QUANTIZATION)")
    A = synth.random_weights(ta, m, k, 81)
    x = synth.random_activations(n, k, 82)
    B = synth.quantize_activations(tb, x)
    G = oracle.f64_gemm(ta, A, tb, B, m, n, k)
    W = gpu.upload_weights(ta, A, m, k)
    C = gpu.mul_mat(W, torch.from_numpy(B).cuda(), tb).cpu().numpy()
    assert not np.isnan(C).any()
    assert rel_err(C, G) <= 2e-6
    ok, O = oracle.sgemm(ta, A, tb, B, m, n, k)
    if ok == 1:
        assert rel_err(C, O) <= 2e-6


@pytest.mark.parametrize("ta,tb", [(T.F16, T.F16), (T.F16, T.F32), (T.BF16, T.BF16), (T.BF16, T.F32)], ids=lambda t: T.NAMES[t])
@pytest.mark.parametrize("shape", [(128, 64, 512), (45, 100, 1024), (200, 130, 768), (33, 9, 256), (300, 70, 2048), (130, 20, 256)], ids=str)
def test_float_types_mfma_gemm(gpu, oracle, ta, tb, shape):
    """F16 / BF16 weights, batches: MFMA (f16 / bf16 inputs, f32 accumulate) straight on the RAW rows; f32 activations
    are converted like ggml does before it calls sgemm (f16 RNE; bf16 nearest-even with NaN quieting).  Criterion as
    above: a double-accumulator GEMM on the converted operands (products are exact in f32 on both sides)."""
    from llamafile_amd import synth
    m, n, k = shape
    A = synth.random_weights(ta, m, k, 91)
    x = synth.random_activations(n, k, 92)
    Bsame = synth.quantize_activations(ta, x)  # the activations in the weight's type
    G = oracle.f64_gemm(ta, A, ta, Bsame, m, n, k)
    W = gpu.upload_weights(ta, A, m, k)
    if tb == T.F32:
        C = gpu.mul_mat(W, torch.from_numpy(x).cuda().view(torch.uint8).view(n, k * 4), T.F32)
    else:
        C = gpu.mul_mat(W, torch.from_numpy(Bsame).cuda(), ta)
    C = C.cpu().numpy()
    assert not np.isnan(C).any()
    assert rel_err(C, G) <= 2e-6
    ok, O = oracle.sgemm(ta, A, ta, Bsame, m, n, k)
    if ok == 1:
        assert rel_err(C, O) <= 2e-6


@pytest.mark.parametrize("t", [T.Q4_K, T.Q5_K, T.Q6_K], ids=lambda t: T.NAMES[t])
@pytest.mark.parametrize("tokens,tasks", [(1, 1), (1, 2), (3, 1), (2, 2)])
@pytest.mark.parametrize("f32in", [False, True], ids=["q8k", "f32"])
def test_mul_mat_id_decode_on_device(gpu, oracle, t, tokens, tasks, f32in):
    """GGML_OP_MUL_MAT_ID for a few tokens: the expert is picked inside the GEMV kernel from the device-resident plan
    (no host read-back).  result[token][thinker] = W[plan[token][thinker]] x thought[token][thinker % tasks]
    (tinyblas_cpu_mixmul.inc:39-50); an out-of-range expert id leaves its result row untouched."""
    from llamafile_amd import synth
    rows, cols, experts, thinkers = 96, 512, 5, 2
    Ws = [synth.random_weights(t, rows, cols, 900 + e) for e in range(experts)]
    packed = torch.cat([gpu.upload_weights(t, W, rows, cols).data for W in Ws])
    x = synth.random_activations(tokens * tasks, cols, 77)
    vdt = T.VEC_DOT[t]
    xq = synth.quantize_activations(vdt, x)
    rng = np.random.default_rng(5)
    plan = rng.integers(0, experts, size=(tokens, thinkers)).astype(np.int32)
    if tokens > 1:
        plan[-1, -1] = experts + 3  # invalid: skipped
    if f32in:
        thought = torch.from_numpy(x).cuda().view(torch.uint8).view(tokens * tasks, cols * 4)
        bt = T.F32
    else:
        thought = torch.from_numpy(xq).cuda()
        bt = vdt
    res = gpu.mul_mat_id(packed, t, rows, cols, experts, thought, bt, tasks, tokens, torch.from_numpy(plan).cuda(), thinkers,
                         prefill=-7.0)
    torch.cuda.synchronize()
    res = res.cpu().numpy()
    for tok in range(tokens):
        for th in range(thinkers):
            ex = int(plan[tok, th])
            if ex >= experts:
                assert (res[tok, th] == -7.0).all()
                continue
            row = tok * tasks + th % tasks
            ok, G = oracle.sgemm(t, Ws[ex], vdt, xq[row:row + 1], rows, 1, cols)
            assert ok == 1
            assert rel_err(res[tok, th], G[0]) <= DEFAULT_TOL, (tok, th)


@pytest.mark.parametrize("t", [T.Q4_K, T.Q5_K, T.Q6_K, T.Q8_0, T.Q4_0], ids=lambda t: T.NAMES[t])
def test_tuned_types_random_shapes(gpu, oracle, t):
    """Seeded sweep over ragged shapes for every type with tuned kernels: rows not a multiple of the row tile, batches
    on both sides of the GEMV/GEMM switch and of the 128-token tile, odd super-block counts, both MFMA bodies."""
    from llamafile_amd import _hip
    rng = np.random.default_rng(1234 + t)
    for case in range(10):
        m = int(rng.integers(1, 300))
        n = int(rng.choice([1, 2, 3, 7, 8, 9, 17, 64, 127, 128, 129, 200]))
        k = 256 * int(rng.integers(1, 6))
        A, B, bt = make_case(t, m, n, k, seed=int(rng.integers(1 << 30)))
        ok, G = oracle.sgemm(t, A, bt, B, m, n, k, nth=3)
        assert ok == 1
        bodies = [0]
        if n > 8 and t in (T.Q4_K, T.Q5_K, T.Q6_K):
            bodies = [_hip.FLAG_GEMM_NARROW, _hip.FLAG_GEMM_WIDE, _hip.FLAG_GEMM_WIDE | _hip.FLAG_GEMM_PLAIN,
                      _hip.FLAG_GEMM_WIDE | _hip.FLAG_PRECISE]
        for body in bodies:
            C = run_gpu(gpu, t, A, B, bt, m, n, k, flags=gpu.host_variant_flags() | body)
            if t == T.Q8_0:
                if n <= 8:  # the vecdot
                    assert np.array_equal(C.view(np.uint32), G.view(np.uint32)), (m, n, k)
                else:  # batches: the f16 MFMA body by default (helpers.q80_batch_tol), the bit-exact kernel on request
                    from helpers import q80_batch_tol
                    assert rel_err(C, G) <= q80_batch_tol(m, k, n), (m, n, k)
                    Cx = run_gpu(gpu, t, A, B, bt, m, n, k, flags=gpu.host_variant_flags() | _hip.FLAG_Q80_EXACT)
                    assert np.array_equal(Cx.view(np.uint32), G.view(np.uint32)), (m, n, k)
            else:
                tol = gemm_tol(t, body, n)
                assert rel_err(C, G) <= tol, (T.NAMES[t], m, n, k, body, rel_err(C, G))


@pytest.mark.parametrize("t", [T.Q4_0, T.Q4_1, T.Q5_0, T.Q5_1], ids=lambda t: T.NAMES[t])
@pytest.mark.parametrize("shape", [(128, 64, 512), (96, 100, 1024), (33, 9, 256), (200, 300, 1280), (64, 130, 768)], ids=str)
@pytest.mark.parametrize("f32in", [False, True], ids=["q8", "f32"])
def test_legacy_types_gemm_vs_oracle(gpu, oracle, t, shape, f32in):
    """Legacy 32-block types, batches: exact integer codes on the MFMA body, f32 block scales per 32 weights (Q4_0 from
    its resident P40 layout, the others from a per-call image); Q4_1 / Q5_1 add m * s with Q8_1 activations."""
    from llamafile_amd import synth
    m, n, k = shape
    A = synth.random_weights(t, m, k, 600 + t)
    x = synth.random_activations(n, k, 601)
    vdt = T.VEC_DOT[t]
    B = synth.quantize_activations(vdt, x)
    ok, G = oracle.sgemm(t, A, vdt, B, m, n, k, nth=2)
    assert ok == 1
    W = gpu.upload_weights(t, A, m, k)
    if f32in:
        C = gpu.mul_mat(W, torch.from_numpy(x).cuda().view(torch.uint8).view(n, k * 4), T.F32)
    else:
        C = gpu.mul_mat(W, torch.from_numpy(B).cuda(), vdt)
    torch.cuda.synchronize()
    assert rel_err(C.cpu().numpy(), G) <= TOL.get(t, DEFAULT_TOL)


@pytest.mark.parametrize("t", [T.Q4_K, T.Q5_K, T.Q6_K], ids=lambda t: T.NAMES[t])
@pytest.mark.parametrize("tokens,tasks", [(5, 1), (37, 2), (150, 1), (200, 2)])
@pytest.mark.parametrize("f32in", [False, True], ids=["q8k", "f32"])
def test_mul_mat_id_batches_grouped_on_device(gpu, oracle, t, tokens, tasks, f32in):
    """GGML_OP_MUL_MAT_ID for batches: rows are grouped by expert ON THE DEVICE (no routing read-back) and all experts run
    in one launch of the 128x128 MFMA body; skewed routing (one hot expert, one unused), an out-of-range id."""
    from llamafile_amd import synth
    rows, cols, experts, thinkers = 160, 512, 6, 2
    Ws = [synth.random_weights(t, rows, cols, 950 + e) for e in range(experts)]
    packed = torch.cat([gpu.upload_weights(t, W, rows, cols).data for W in Ws])
    x = synth.random_activations(tokens * tasks, cols, 78)
    xq = synth.quantize_activations(T.Q8_K, x)
    rng = np.random.default_rng(6)
    plan = rng.choice([0, 0, 0, 1, 2, 4, 5], size=(tokens, thinkers)).astype(np.int32)  # expert 3 never used, 0 hot
    plan[-1, -1] = experts + 2  # invalid: row left untouched
    if f32in:
        thought = torch.from_numpy(x).cuda().view(torch.uint8).view(tokens * tasks, cols * 4)
        bt = T.F32
    else:
        thought = torch.from_numpy(xq).cuda()
        bt = T.Q8_K
    from llamafile_amd import _hip
    golden = {}
    # batches of K-quant experts: scaled operands by default, exact integer codes with LFAMD_FLAG_PRECISE
    for flags, tol in ((0, SCALED_TOL if tokens > 4 else GEMM_TOL[t]), (_hip.FLAG_PRECISE, GEMM_TOL[t])):
        res = gpu.mul_mat_id(packed, t, rows, cols, experts, thought, bt, tasks, tokens, torch.from_numpy(plan).cuda(), thinkers,
                             flags=gpu.host_variant_flags() | flags, prefill=-7.0)
        torch.cuda.synchronize()
        res = res.cpu().numpy()
        for ex in range(experts):
            sel = [(tok, th) for tok in range(tokens) for th in range(thinkers) if plan[tok, th] == ex]
            if not sel:
                continue
            if ex not in golden:
                Bx = np.stack([xq[tok * tasks + th % tasks] for tok, th in sel])
                ok, golden[ex] = oracle.sgemm(t, Ws[ex], T.Q8_K, Bx, rows, len(sel), cols)
                assert ok == 1
            got = np.stack([res[tok, th] for tok, th in sel])
            assert rel_err(got, golden[ex]) <= tol, (ex, flags, rel_err(got, golden[ex]))
        assert (res[-1, -1] == -7.0).all()


def test_mul_mat_id_routing_many_experts(gpu):
    """Device-side routing with many experts and a ragged row count (40 experts, 171 tokens x 3 thinkers = 513 rows: three
    blocks of the ordered fill, the last with one row): every result row equals its expert's row for its
    activation row in the plain batch mat-mul of that expert (exact integer codes on both sides: LFAMD_FLAG_PRECISE), rows with an out-of-range id stay untouched."""
    from llamafile_amd import synth, _hip
    t, rows, cols, experts, thinkers, tokens, tasks = T.Q4_K, 64, 512, 40, 3, 171, 3
    Ws = [gpu.upload_weights(t, synth.random_weights(t, rows, cols, 1200 + e), rows, cols) for e in range(experts)]
    packed = torch.cat([W.data for W in Ws])
    x = synth.random_activations(tokens * tasks, cols, 79)
    xq = torch.from_numpy(synth.quantize_activations(T.Q8_K, x)).cuda()
    rng = np.random.default_rng(8)
    plan = rng.integers(0, experts, size=(tokens, thinkers)).astype(np.int32)
    plan[5, 1] = -1
    plan[170, 2] = experts
    res = gpu.mul_mat_id(packed, t, rows, cols, experts, xq, T.Q8_K, tasks, tokens, torch.from_numpy(plan).cuda(), thinkers,
                         flags=gpu.host_variant_flags() | _hip.FLAG_PRECISE, prefill=-7.0).cpu().numpy()
    assert (res[5, 1] == -7.0).all() and (res[170, 2] == -7.0).all()
    fl = gpu.host_variant_flags() | _hip.FLAG_PRECISE
    per_expert = {e: gpu.mul_mat(Ws[e], xq, T.Q8_K, flags=fl).cpu().numpy() for e in range(experts)}  # [tokens*tasks, rows]
    worst = 0.0
    for tok in range(tokens):
        for th in range(thinkers):
            e = int(plan[tok, th])
            if 0 <= e < experts:
                worst = max(worst, rel_err(res[tok, th], per_expert[e][tok * tasks + th % tasks]))
    assert worst <= GEMM_TOL[t], worst


@pytest.mark.parametrize("t", [T.Q2_K, T.Q3_K, T.IQ4_XS, T.Q5_1, T.Q4_0, T.Q8_0], ids=lambda t: T.NAMES[t])
@pytest.mark.parametrize("tokens", [1, 19])
def test_mul_mat_id_other_expert_types(gpu, oracle, t, tokens):
    """Expert stacks of the types without a grouped kernel (gather + one mat-mul per expert, like the reference's GPU path,
    ggml-cuda.cu.patch:18499-18635): the stack is the experts' PACKED images back to back (lfamd_packed_size apart — since
    round 2 that is not the GGUF size for Q2_K / Q3_K / IQ4_XS / Q4_1 / Q5_0 / Q5_1); every row against the oracle."""
    from llamafile_amd import synth
    rows, cols, experts, thinkers, tasks = 72, 512, 5, 2, 1
    raws = [synth.random_weights(t, rows, cols, 1300 + e) for e in range(experts)]
    packed = torch.cat([gpu.upload_weights(t, W, rows, cols).data for W in raws])
    x = synth.random_activations(tokens * tasks, cols, 81)
    vdt = T.VEC_DOT[t]
    xq = synth.quantize_activations(vdt, x)
    rng = np.random.default_rng(9)
    plan = np.stack([rng.permutation(experts)[:thinkers] for _ in range(tokens)]).astype(np.int32)
    res = gpu.mul_mat_id(packed, t, rows, cols, experts, torch.from_numpy(xq).cuda(), vdt, tasks, tokens,
                         torch.from_numpy(plan).cuda(), thinkers, prefill=-7.0).cpu().numpy()
    for tok in range(tokens):
        for th in range(thinkers):
            e = int(plan[tok, th])
            ok, G = oracle.sgemm(t, raws[e], vdt, xq[tok * tasks + th % tasks:tok * tasks + th % tasks + 1], rows, 1, cols, nth=1)
            assert ok == 1
            if t == T.Q8_0:
                from helpers import q80_batch_tol
                assert rel_err(res[tok, th], G[0]) <= (1e-3 if tokens > 1 else 2e-6)
            else:  # (an expert with more than 8 rows runs the MFMA body: IQ4_XS rounds |sc * v| above 2048 to f16 there)
                tol = 1e-3 if t == T.IQ4_XS and tokens > 1 else TOL.get(t, DEFAULT_TOL)
                assert rel_err(res[tok, th], G[0]) <= tol, (T.NAMES[t], tok, th)


@pytest.mark.gpu
@pytest.mark.parametrize("tokens,thinkers,count", [(1, 2, 2), (3, 2, 2), (1, 3, 2), (2, 2, 3), (9, 2, 2)], ids=str)
@pytest.mark.parametrize("t", [T.Q4_K, T.Q6_K], ids=lambda t: T.NAMES[t])
def test_mul_mat_id_multi_equals_separate_calls(gpu, t, tokens, thinkers, count):
    """lfamd_mul_mat_id_multi: ffn_gate_exps + ffn_up_exps (same activations, same routing) share decode launches of up to four
    (tensor, thinker) GEMVs; the results are the bits of one lfamd_mul_mat_id per tensor (itself oracle-tested above), also where
    it falls back to those calls (9 tokens), also with an out-of-range expert id."""
    import ctypes as C
    from llamafile_amd import synth, _hip
    rows, cols, experts = 80, 768, 5
    stacks = [torch.cat([gpu.upload_weights(t, synth.random_weights(t, rows, cols, 1200 + 10 * j + e), rows, cols).data for e in range(experts)])
              for j in range(count)]
    x = synth.random_activations(tokens, cols, 79)
    thought = torch.from_numpy(x).cuda().view(torch.uint8).view(tokens, cols * 4)
    rng = np.random.default_rng(8)
    plan = rng.integers(0, experts, size=(tokens, thinkers)).astype(np.int32)
    if tokens > 1:
        plan[-1, 0] = experts + 1
    pd = torch.from_numpy(plan).cuda()
    want = [gpu.mul_mat_id(s, t, rows, cols, experts, thought, T.F32, 1, tokens, pd, thinkers, prefill=-3.0) for s in stacks]
    got = [torch.full((tokens, thinkers, rows), -3.0, device="cuda") for _ in range(count)]
    L = _hip.lib()
    need = L.lfamd_mul_mat_id_workspace(t, rows, cols, experts, tokens, thinkers)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
    wp = (C.c_void_p * count)(*[s.data_ptr() for s in stacks])
    rp = (C.c_void_p * count)(*[g.data_ptr() for g in got])
    rc = L.lfamd_mul_mat_id_multi(t, count, wp, rows, cols, experts, T.F32, C.c_void_p(thought.data_ptr()), thought.stride(0), 1, tokens,
                                  C.c_void_p(pd.data_ptr()), thinkers, rp, C.c_void_p(ws.data_ptr()), ws.numel(), gpu.host_variant_flags(),
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, L.lfamd_last_error()
    torch.cuda.synchronize()
    for g, w in zip(got, want):
        assert torch.equal(g.view(torch.int32), w.view(torch.int32))
