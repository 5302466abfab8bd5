"""Decode (n <= 8) on float weights (F16 / BF16 / F32 rows in RAW layout) through the device C ABI: gemv_float.hip against
the float64 product of the same operands.  Reference: tinyBLAS's float instantiations (tinyblas_cpu_sgemm.inc:45-150):
weights and activations widened to f32, f32 fma — so the only legitimate difference is the order of the f32 additions.
Shapes cover an odd row count (the second row of the last pair is absent), rows that are not a whole group of chunks
(zero-padded LDS image + descriptor-clipped loads), several column counts incl. the column split of deep rows, and
activations given in the weight type (tinyBLAS accepts F16 x F16 / BF16 x BF16)."""
import numpy as np
import pytest
import torch

from llamafile_amd import ggml_types as T
from helpers import rel_err

pytestmark = pytest.mark.gpu

TORCH = {T.F32: torch.float32, T.F16: torch.float16, T.BF16: torch.bfloat16}
SHAPES = [(33, 1000, 1), (256, 4096, 1), (257, 4096, 3), (64, 11008, 2), (31, 14336, 8), (2, 8, 1), (1, 264, 5), (4099, 2048, 4)]


@pytest.mark.parametrize("same_type_b", [False, True], ids=["xf32", "xsame"])
@pytest.mark.parametrize("t", [T.F16, T.BF16, T.F32], ids=lambda t: T.NAMES[t])
@pytest.mark.parametrize("m,k,n", SHAPES, ids=lambda v: str(v))
def test_float_decode_matches_f64_product(gpu, t, m, k, n, same_type_b):
    if same_type_b and t == T.F32:
        pytest.skip("same as xf32")
    g = torch.Generator(device="cuda")
    g.manual_seed(m * 31 + k + n)
    w = (torch.rand((m, k), device="cuda", generator=g) * 2 - 1).to(TORCH[t])
    x = (torch.rand((n, k), device="cuda", generator=g) * 2 - 1)
    bt = t if same_type_b else T.F32
    xb = x.to(TORCH[bt])
    W = gpu.upload_weights(t, w.view(torch.uint8).reshape(m, -1), m, k)
    out = gpu.mul_mat(W, xb.contiguous().view(torch.uint8).reshape(n, -1), bt, n=n).cpu().numpy()
    ref = (xb.double() @ w.double().T).cpu().numpy()
    assert out.shape == (n, m)
    assert rel_err(out, ref) <= 2e-6  # f32 accumulation of <= 14336 products of magnitude <= 1
    # every output, not just the largest: |err| <= 1e-5 * sum |w x| bound
    bound = (xb.double().abs() @ w.double().abs().T).cpu().numpy()
    assert np.all(np.abs(out - ref) <= 1e-5 * bound + 1e-30)


def test_float_decode_unaligned_activation_stride_falls_back(gpu):
    """Activation rows whose stride is not a 16-byte multiple cannot take the 16-byte loads of gemv_float: the dispatcher
    hands them to the generic kernel — same result."""
    m, k, n = 40, 512, 3
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    w = (torch.rand((m, k), device="cuda", generator=g) * 2 - 1).to(torch.float16)
    x = torch.rand((n, k), device="cuda", generator=g) * 2 - 1
    W = gpu.upload_weights(T.F16, w.view(torch.uint8).reshape(m, -1), m, k)
    buf = torch.zeros((n, k * 4 + 4), dtype=torch.uint8, device="cuda")
    buf[:, :k * 4] = x.view(torch.uint8).reshape(n, k * 4)
    out = gpu.mul_mat(W, buf, T.F32, n=n).cpu().numpy()
    ref = (x.double() @ w.double().T).cpu().numpy()
    assert rel_err(out, ref) <= 2e-6
