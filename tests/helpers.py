"""Shared helpers for the parity tests."""
import numpy as np

from llamafile_amd import ggml_types as T, synth


def make_case(t, m, n, k, seed):
    """Random weights of type t [m, k] + activations quantised to t's vec_dot format [n, k]."""
    A = synth.random_weights(t, m, k, seed)
    x = synth.random_activations(n, k, seed + 1)
    bt = T.VEC_DOT[t]
    B = synth.quantize_activations(bt, x)
    return A, B, bt


def rel_err(C, G):
    """max |C-G| / max |G| (normwise, like the logits tolerance of the north star)."""
    G = np.asarray(G, dtype=np.float64)
    C = np.asarray(C, dtype=np.float64)
    return float(np.abs(C - G).max() / max(np.abs(G).max(), 1e-30))


def elem_err(C, G, rtol=1e-3):
    """Element-wise statistic beside the normwise rel_err: the fraction of outputs with
    |C - G| > rtol * |G| + rtol * rms(G), and the largest |C - G| / (|G| + rms(G)).  A normwise 1e-3 can hide small
    outputs that are off by far more; this one cannot."""
    G = np.asarray(G, dtype=np.float64)
    C = np.asarray(C, dtype=np.float64)
    rms = float(np.sqrt(np.mean(G * G))) or 1e-30
    d = np.abs(C - G)
    return float(np.mean(d > rtol * np.abs(G) + rtol * rms)), float((d / (np.abs(G) + rms)).max())


def q80_batch_tol(m=128, k=128, n=16, flags=0):
    """Q8_0 batches (n > 8): what the call runs is asked of the module (lfamd_mul_mat_is_exact).  Default flags, rows of whole
    128-weight quads: the f16 MFMA body on the resident P80 image (gemm_lf.hip: f16(d * q) x f16(d8 * code), one f16 rounding per
    operand) — the north star's tolerance for f16 MFMA paths, 1e-3 (measured 2-4e-4); the same for a process that opted into the
    vendor GEMM (LFAMD_USE_BLASLT=1).  Other row lengths, LFAMD_FLAG_PRECISE / LFAMD_FLAG_Q80_EXACT: the bit-exact kernel."""
    from llamafile_amd import _hip, ggml_types as T
    return 2e-6 if _hip.lib().lfamd_mul_mat_is_exact(T.Q8_0, m, k, n, flags) else 1e-3
