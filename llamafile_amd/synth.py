"""Synthetic inputs for the quantized-matmul path (no GGUF file exists offline).

Weights are random block bytes with finite f16 scales, activations are
U(-1,1) like the reference's ``numba()`` (/root/reference/llamafile/numba.h:21-27),
then quantised to the activation format the weight type multiplies with
(SURVEY.md §8d).  Everything is seeded and numpy-only.
"""
from __future__ import annotations

import numpy as np

from . import ggml_types as T

# byte offset of the f16 scale fields inside one block, per type: (d_off, second_off or None)
_SCALE_OFF = {
    T.Q4_0: (0, None), T.Q4_1: (0, 2), T.Q5_0: (0, None), T.Q5_1: (0, 2), T.Q8_0: (0, None),
    T.Q2_K: (80, 82), T.Q3_K: (108, None), T.Q4_K: (0, 2), T.Q5_K: (0, 2), T.Q6_K: (208, None),
    T.IQ4_XS: (0, None),
}


def _f16_bytes(vals: np.ndarray) -> np.ndarray:
    return vals.astype(np.float16).view(np.uint8).reshape(vals.shape + (2,))


def random_weights(t: int, rows: int, cols: int, seed: int) -> np.ndarray:
    """Random tensor of ggml type ``t``: uint8 array [rows, row_size(t, cols)] (raw GGUF layout).

    Quant codes are uniform random bytes; ``d`` is log-uniform in [2^-10, 2^-6] and the second
    scale (dmin / m) uniform in [0, 2^-7], so every value is finite."""
    rng = np.random.default_rng(seed)
    if t == T.F32:
        return (rng.random((rows, cols), dtype=np.float32) * 2 - 1).view(np.uint8).reshape(rows, cols * 4)
    if t == T.F16:
        return (rng.random((rows, cols), dtype=np.float32) * 2 - 1).astype(np.float16).view(np.uint8).reshape(rows, cols * 2)
    if t == T.BF16:
        f = rng.random((rows, cols), dtype=np.float32) * 2 - 1
        return f32_to_bf16(f).view(np.uint8).reshape(rows, cols * 2)
    nb = cols // T.BLCK[t]
    ts = T.TYPE_SIZE[t]
    raw = rng.integers(0, 256, size=(rows, nb, ts), dtype=np.uint8)
    d_off, s_off = _SCALE_OFF[t]
    d = np.exp2(rng.uniform(-10, -6, size=(rows, nb))).astype(np.float32)
    raw[:, :, d_off:d_off + 2] = _f16_bytes(d)
    if s_off is not None:
        s = rng.uniform(0, 2.0 ** -7, size=(rows, nb)).astype(np.float32)
        raw[:, :, s_off:s_off + 2] = _f16_bytes(s)
    if t == T.Q8_0:
        q = raw[:, :, 2:].view(np.int8)
        q[q == -128] = -127  # the Q8_0 quantiser never emits -128
    return raw.reshape(rows, nb * ts)


def rescale_blocks(t: int, raw: np.ndarray, factor: float) -> None:
    """Multiply the f16 block scales (d, and dmin / m where the type has one) of a raw tensor in place — e.g. 2^-5 moves
    the synthetic d range 2^-10..2^-6 to the 2^-15..2^-11 of real K-quant files, where d * sc of a small sub-block
    scale is an f16 subnormal."""
    nb = raw.shape[1] // T.TYPE_SIZE[t]
    blk = raw.reshape(raw.shape[0], nb, T.TYPE_SIZE[t])
    d_off, s_off = _SCALE_OFF[t]
    for off in (d_off, s_off):
        if off is None:
            continue
        v = np.ascontiguousarray(blk[:, :, off:off + 2]).view(np.float16).astype(np.float32)[..., 0] * np.float32(factor)
        blk[:, :, off:off + 2] = _f16_bytes(v)


def random_activations(rows: int, cols: int, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return (rng.random((rows, cols), dtype=np.float32) * 2 - 1).astype(np.float32)


# ---------------------------------------------------------------------------------------------
# activation quantisers (host side).  Bit-identical to upstream's scalar reference quantisers as
# restated in oracle/oracle.c — tests check that; the product only needs them to feed the GPU.


def f32_to_bf16(f: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(f, dtype=np.float32).view(np.uint32)
    nan = (x & 0x7FFFFFFF) > 0x7F800000
    r = ((x + (0x7FFF + ((x >> 16) & 1))) >> 16).astype(np.uint16)
    r[nan] = ((x[nan] >> 16) | 64).astype(np.uint16)
    return r


def _roundf(x: np.ndarray) -> np.ndarray:
    """C roundf (half away from zero) on float32 without double rounding."""
    r = np.trunc(x)
    frac = np.abs(x - r)
    return r + np.sign(x) * (frac >= 0.5)


def quantize_q8_0(x: np.ndarray) -> np.ndarray:
    """f32 [n, k] -> block_q8_0 bytes [n, k/32*34]."""
    n, k = x.shape
    xb = x.reshape(n, k // 32, 32).astype(np.float32)
    amax = np.abs(xb).max(axis=2)
    d = (amax / np.float32(127.0)).astype(np.float32)
    with np.errstate(divide="ignore"):
        idv = np.where(d != 0, np.float32(1.0) / d, np.float32(0.0)).astype(np.float32)
    q = _roundf((xb * idv[:, :, None]).astype(np.float32)).astype(np.int8)
    out = np.empty((n, k // 32, 34), dtype=np.uint8)
    out[:, :, 0:2] = _f16_bytes(d)
    out[:, :, 2:] = q.view(np.uint8)
    return out.reshape(n, -1)


def quantize_q8_1(x: np.ndarray) -> np.ndarray:
    """f32 [n, k] -> block_q8_1 bytes [n, k/32*36]; s = d * sum(qs)."""
    n, k = x.shape
    xb = x.reshape(n, k // 32, 32).astype(np.float32)
    amax = np.abs(xb).max(axis=2)
    d = (amax / np.float32(127.0)).astype(np.float32)
    with np.errstate(divide="ignore"):
        idv = np.where(d != 0, np.float32(1.0) / d, np.float32(0.0)).astype(np.float32)
    q = _roundf((xb * idv[:, :, None]).astype(np.float32)).astype(np.int8)
    s = (q.astype(np.int32).sum(axis=2).astype(np.float32) * d).astype(np.float32)
    out = np.empty((n, k // 32, 36), dtype=np.uint8)
    out[:, :, 0:2] = _f16_bytes(d)
    out[:, :, 2:4] = _f16_bytes(s)
    out[:, :, 4:] = q.view(np.uint8)
    return out.reshape(n, -1)


def quantize_q8_K(x: np.ndarray) -> np.ndarray:
    """f32 [n, k] -> llamafile-order block_q8_K bytes [n, k/256*292] = {f32 d, i16 bsums[16], i8 qs[256]}
    (/root/reference/llama.cpp.patches/patches/ggml-common.h.patch:25-35)."""
    n, k = x.shape
    xb = x.reshape(n, k // 256, 256).astype(np.float32)
    idx = np.abs(xb).argmax(axis=2)
    mx = np.take_along_axis(xb, idx[:, :, None], axis=2)[:, :, 0]
    zero = mx == 0
    with np.errstate(divide="ignore"):
        iscale = np.where(zero, np.float32(0), np.float32(-128.0) / np.where(zero, np.float32(1), mx)).astype(np.float32)
    v = np.rint((iscale[:, :, None] * xb).astype(np.float32))
    q = np.minimum(v, 127).astype(np.int8)
    with np.errstate(divide="ignore"):
        d = np.where(zero, np.float32(0), np.float32(1.0) / np.where(zero, np.float32(1), iscale)).astype(np.float32)
    bsums = q.reshape(n, k // 256, 16, 16).astype(np.int32).sum(axis=3).astype(np.int16)
    out = np.empty((n, k // 256, 292), dtype=np.uint8)
    out[:, :, 0:4] = d.view(np.uint8).reshape(n, k // 256, 4)
    out[:, :, 4:36] = bsums.view(np.uint8).reshape(n, k // 256, 32)
    out[:, :, 36:] = q.view(np.uint8)
    return out.reshape(n, -1)


def quantize_activations(vec_dot_type: int, x: np.ndarray) -> np.ndarray:
    if vec_dot_type == T.Q8_0:
        return quantize_q8_0(x)
    if vec_dot_type == T.Q8_1:
        return quantize_q8_1(x)
    if vec_dot_type == T.Q8_K:
        return quantize_q8_K(x)
    if vec_dot_type == T.F32:
        return np.ascontiguousarray(x, dtype=np.float32).view(np.uint8).reshape(x.shape[0], -1)
    if vec_dot_type == T.F16:
        return x.astype(np.float16).view(np.uint8).reshape(x.shape[0], -1)
    if vec_dot_type == T.BF16:
        return f32_to_bf16(x).view(np.uint8).reshape(x.shape[0], -1)
    raise ValueError(vec_dot_type)


def random_weights_torch(t: int, rows: int, cols: int, seed: int, device="cuda"):
    """Same recipe as random_weights() but generated on the device (bench-sized tensors: GBs).
    Returns a torch uint8 tensor [rows, row_size]."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if t in (T.F32, T.F16, T.BF16):  # float weights: uniform in [-1, 1)
        w = torch.rand((rows, cols), device=device, generator=g) * 2.0 - 1.0
        w = w.to({T.F32: torch.float32, T.F16: torch.float16, T.BF16: torch.bfloat16}[t])
        return w.view(torch.uint8).reshape(rows, -1)
    nb = cols // T.BLCK[t]
    ts = T.TYPE_SIZE[t]
    raw = torch.randint(0, 256, (rows, nb, ts), dtype=torch.uint8, device=device, generator=g)
    d_off, s_off = _SCALE_OFF[t]
    d = torch.exp2(torch.rand((rows, nb), device=device, generator=g) * 4.0 - 10.0).to(torch.float16)
    raw[:, :, d_off:d_off + 2] = d.view(torch.uint8).reshape(rows, nb, 2)
    if s_off is not None:
        s = (torch.rand((rows, nb), device=device, generator=g) * (2.0 ** -7)).to(torch.float16)
        raw[:, :, s_off:s_off + 2] = s.view(torch.uint8).reshape(rows, nb, 2)
    if t == T.Q8_0:
        q = raw[:, :, 2:].view(torch.int8)
        q[q == -128] = -127
    return raw.reshape(rows, nb * ts)
