"""The loader-wave batch bodies of csrc/gemm_lf.hip against the oracle.
  * Q8_0 (rows of whole 128-weight quads): f16(d * q) built in registers from the ONE resident P80 image x f16(d8 * q8) — the
    arithmetic of the reference's GPU path for such a batch (tinyblas.cu:142-226: dequantise, f16 GEMM), <= 1e-3 normwise of the
    CPU reference (tinyblas_cpu.h:934-971 restated in the oracle) and no element beyond 3e-3 (|G| + rms); both tile widths, f32
    and Q8_0 input, ragged rows / tokens, one to many quads, ldc > m, sibling matrices in one launch, and the bit-exact kernel
    behind LFAMD_FLAG_Q80_EXACT and for rows that are not whole quads.
  * F16 / BF16: the RAW rows through LDS-DMA, f32 accumulate; 2e-6 of a double-accumulator GEMM on the converted operands."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from llamafile_amd import _hip, ggml_types as T, synth
from helpers import elem_err, rel_err

pytestmark = pytest.mark.gpu

# (m, n, k): one quad, three quads, many; fewer rows than a P80 tile, rows that are no multiple of 8 / 32 / 128; tokens around the
# 64- and 128-token tile edges; (1100, 700, 512) does not fit one round of 128 x 64 tiles and takes the wide tile by default, and so
# do (4224, 520, 128) and (4300, 513, 256): its ring of three stages with one and two stages to go round (two barriers per stage)
SHAPES = [(5, 9, 128), (64, 20, 128), (40, 33, 384), (129, 65, 256), (300, 128, 1024), (1000, 129, 640), (1100, 700, 512), (2048, 64, 4096),
          (4224, 520, 128), (4300, 513, 256)]


@pytest.mark.parametrize("shape", SHAPES, ids=str)
@pytest.mark.parametrize("f32in", [False, True], ids=["q80", "f32"])
def test_q8_0_batch_body_vs_oracle(gpu, oracle, shape, f32in):
    m, n, k = shape
    A = synth.random_weights(T.Q8_0, m, k, 131)
    x = synth.random_activations(n, k, 132)
    B = synth.quantize_activations(T.Q8_0, x)
    assert _hip.lib().lfamd_mul_mat_is_exact(T.Q8_0, m, k, n, gpu.host_variant_flags()) == 0  # (the default route: the f16 MFMA body)
    v = oracle.variant("zen4" if gpu.host_variant_flags() & _hip.FLAG_Q0_VREGS32 else "avx2")
    ok, G = oracle.sgemm(T.Q8_0, A, T.Q8_0, B, m, n, k, nth=8, v=v)
    assert ok == 1
    W = gpu.upload_weights(T.Q8_0, A, m, k)
    assert W.data.numel() == (m + 7) // 8 * (k // 128) * 1088 + (-((m + 7) // 8 * (k // 128) * 1088) % 256)  # ONE image: P80
    Bd = torch.from_numpy(x).cuda().view(torch.uint8).view(n, k * 4) if f32in else torch.from_numpy(B).cuda()
    Cd = gpu.mul_mat(W, Bd, T.F32 if f32in else T.Q8_0, n=n).cpu().numpy()
    assert not np.isnan(Cd).any()
    assert rel_err(Cd, G) <= 1e-3, rel_err(Cd, G)
    frac, worst = elem_err(Cd, G, rtol=3e-3)
    assert frac == 0.0, (frac, worst)
    # the same call with the bit-exact kernel
    Cx = gpu.mul_mat(W, Bd, T.F32 if f32in else T.Q8_0, n=n, flags=gpu.host_variant_flags() | _hip.FLAG_Q80_EXACT).cpu().numpy()
    assert np.array_equal(Cx.view(np.uint32), G.view(np.uint32))


def test_q8_0_rows_that_are_not_whole_quads_run_the_exact_kernel(gpu, oracle):
    m, n, k = 72, 40, 160
    A = synth.random_weights(T.Q8_0, m, k, 141)
    B = synth.quantize_activations(T.Q8_0, synth.random_activations(n, k, 142))
    assert _hip.lib().lfamd_mul_mat_is_exact(T.Q8_0, m, k, n, gpu.host_variant_flags()) == 1
    v = oracle.variant("zen4" if gpu.host_variant_flags() & _hip.FLAG_Q0_VREGS32 else "avx2")
    ok, G = oracle.sgemm(T.Q8_0, A, T.Q8_0, B, m, n, k, nth=2, v=v)
    assert ok == 1
    Cd = gpu.mul_mat(gpu.upload_weights(T.Q8_0, A, m, k), torch.from_numpy(B).cuda(), T.Q8_0, n=n).cpu().numpy()
    assert np.array_equal(Cd.view(np.uint32), G.view(np.uint32))


def test_q8_0_both_tile_widths_agree(gpu):
    """LFAMD_LF_NT=2 / 4 (child processes): the 64- and 128-token tiles sum a row's quads in the same order — the same bits."""
    code = ("import sys, numpy as np, torch; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')\n"
            "from llamafile_amd import sgemm, synth, ggml_types as T\n"
            "sgemm.init(0); m, n, k = 700, 300, 768\n"
            "W = sgemm.upload_weights(T.Q8_0, synth.random_weights(T.Q8_0, m, k, 151), m, k)\n"
            "x = torch.from_numpy(synth.random_activations(n, k, 152)).cuda()\n"
            "c = sgemm.mul_mat(W, x.view(torch.uint8).view(n, k * 4), T.F32, n=n).cpu().numpy()\n"
            "sys.stdout.buffer.write(c.tobytes())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for nt in ("2", "4"):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env={**os.environ, "LFAMD_LF_NT": nt}, capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout)
    assert len(outs[0]) == 700 * 300 * 4 and outs[0] == outs[1]


def test_q8_0_batch_body_respects_ldc_and_leaves_the_rest_alone(gpu):
    m, n, k, ldc = 520, 150, 256, 520 + 24
    W = gpu.upload_weights(T.Q8_0, synth.random_weights(T.Q8_0, m, k, 161), m, k)
    Bd = torch.from_numpy(synth.random_activations(n, k, 162)).cuda().view(torch.uint8).view(n, k * 4)
    ref = gpu.mul_mat(W, Bd, T.F32, n=n)
    out = torch.full((n, ldc), 7.0, device="cuda")
    L = _hip.lib()
    ws = torch.empty(gpu.workspace_bytes(T.Q8_0, m, k, n), dtype=torch.uint8, device="cuda")
    rc = L.lfamd_mul_mat(T.Q8_0, C.c_void_p(W.data.data_ptr()), m, k, T.F32, C.c_void_p(Bd.data_ptr()), k * 4, n, C.c_void_p(out.data_ptr()),
                         ldc, C.c_void_p(ws.data_ptr()), ws.numel(), gpu.host_variant_flags(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, L.lfamd_last_error()
    torch.cuda.synchronize()
    assert torch.equal(out[:, :m], ref) and bool((out[:, m:] == 7.0).all())


@pytest.mark.parametrize("count", [2, 3, 4])
def test_q8_0_sibling_matrices_share_the_staging(gpu, count):
    """lfamd_mul_mat_multi on Q8_0 batches: one staging of the activations, one launch over the concatenated row blocks — the bits
    of separate calls (a work-group's tile does not depend on which launch it belongs to)."""
    k, n = 512, 90
    ms = [300, 40, 7, 1030][:count]
    Ws = [gpu.upload_weights(T.Q8_0, synth.random_weights(T.Q8_0, m, k, 170 + i), m, k) for i, m in enumerate(ms)]
    x = torch.from_numpy(synth.random_activations(n, k, 175)).cuda()
    fused = gpu.mul_mat_multi(Ws, x.view(torch.uint8), T.F32, n=n)
    for W, f in zip(Ws, fused):
        sep = gpu.mul_mat(W, x.view(torch.uint8), T.F32, n=n)
        assert torch.equal(f.view(torch.int32), sep.view(torch.int32))


@pytest.mark.parametrize("ta", [T.F16, T.BF16], ids=lambda t: T.NAMES[t])
@pytest.mark.parametrize("shape", [(5, 9, 256), (130, 65, 256), (300, 128, 768), (1000, 200, 1024), (2048, 64, 4096)], ids=str)
@pytest.mark.parametrize("f32in", [False, True], ids=["same", "f32"])
def test_float_batch_body_vs_f64(gpu, oracle, ta, shape, f32in):
    m, n, k = shape
    A = synth.random_weights(ta, m, k, 181)
    x = synth.random_activations(n, k, 182)
    Bsame = synth.quantize_activations(ta, x)  # the activations in the weight's type (what ggml converts to before sgemm)
    G = oracle.f64_gemm(ta, A, ta, Bsame, m, n, k)
    W = gpu.upload_weights(ta, A, m, k)
    if f32in:
        Cd = gpu.mul_mat(W, torch.from_numpy(x).cuda().view(torch.uint8).view(n, k * 4), T.F32, n=n)
    else:
        Cd = gpu.mul_mat(W, torch.from_numpy(Bsame).cuda(), ta, n=n)
    Cd = Cd.cpu().numpy()
    assert not np.isnan(Cd).any()
    assert rel_err(Cd, G) <= 2e-6, rel_err(Cd, G)
    # the 128 x 128 wide body of the earlier rounds (testing flag): the same products, another summation order
    Cw = gpu.mul_mat(W, torch.from_numpy(Bsame).cuda(), ta, n=n, flags=gpu.host_variant_flags() | _hip.FLAG_GEMM_WIDE).cpu().numpy()
    assert rel_err(Cd, Cw) <= 2e-6
