"""The device pack kernels produce exactly the layout lfamd_device.h documents (numpy restatement)."""
import numpy as np
import pytest

from llamafile_amd import ggml_types as T, synth
import pack_ref

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("t,ref", [(T.Q4_K, pack_ref.pack_q4k), (T.Q6_K, pack_ref.pack_q6k), (T.Q8_0, pack_ref.pack_q80)],
                         ids=["Q4_K", "Q6_K", "Q8_0"])
@pytest.mark.parametrize("shape", [(64, 512), (37, 1024), (7, 256)], ids=str)
def test_pack_matches_layout_spec(gpu, t, ref, shape):
    rows, cols = shape
    raw = synth.random_weights(t, rows, cols, seed=11)
    W = gpu.upload_weights(t, raw, rows, cols)
    got = W.data.cpu().numpy()
    want = ref(raw, rows, cols)
    if t == T.Q8_0:  # ONE image, P80 (vecdot, exact batches, the f16 MFMA batch body); a process that opted into the vendor GEMM
        from llamafile_amd import _hip  # (LFAMD_USE_BLASLT=1) keeps f16(d * q) rows behind it, 256-aligned
        assert np.array_equal(got[: want.size], want)
        if _hip.lib().lfamd_vendor_gemm_available():
            first = (want.size + 255) // 256 * 256
            blk = raw.reshape(rows, cols // 32, 34)
            d = blk[:, :, :2].copy().view(np.float16).astype(np.float32)  # [rows, nblk, 1]
            q = blk[:, :, 2:].view(np.int8).astype(np.float32)
            second = (d * q).astype(np.float16).reshape(-1).view(np.uint8)
            assert got.size == first + second.size
            assert np.array_equal(got[first:], second)
        else:
            assert got.size == (want.size + 255) // 256 * 256
        return
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_pack_honours_raw_row_stride(gpu):
    rows, cols = 40, 512
    raw = synth.random_weights(T.Q4_K, rows, cols, seed=3)
    padded = np.zeros((rows, raw.shape[1] + 48), dtype=np.uint8)
    padded[:, : raw.shape[1]] = raw
    a = gpu.upload_weights(T.Q4_K, raw, rows, cols).data.cpu().numpy()
    b = gpu.upload_weights(T.Q4_K, padded, rows, cols).data.cpu().numpy()
    assert np.array_equal(a, b)


@pytest.mark.parametrize("t", [T.Q2_K, T.Q3_K], ids=lambda t: T.NAMES[t])
@pytest.mark.parametrize("shape", [(64, 512), (37, 1024), (7, 256)], ids=str)
def test_compact_images_expand_to_the_canonical_image(gpu, t, shape):
    """Q2_K / Q3_K are resident as compact images (two K-steps per code dword, Q3_K's third bit on its own lattice: PK2 / PK3 in
    lfamd_device.h); batches expand them per call into the canonical PCK image the MFMA body reads.  pack + expand must give
    exactly the image the canonical builder makes from the GGUF rows (what was resident until round 2)."""
    import ctypes as C
    import torch
    from llamafile_amd import _hip
    rows, cols = shape
    raw = synth.random_weights(t, rows, cols, seed=17)
    W = gpu.upload_weights(t, raw, rows, cols)
    L = C.CDLL(_hip.HIP_SO)
    L.lfamd_wprep16_bytes.restype = C.c_size_t
    nbytes = L.lfamd_wprep16_bytes(C.c_long(rows), C.c_long(cols))
    assert W.data.numel() < 0.75 * nbytes  # (the point of the exercise)
    rawd = torch.from_numpy(raw).cuda()
    want = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    got = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.lfamd_launch_wprep16(t, C.c_void_p(rawd.data_ptr()), C.c_size_t(raw.shape[1]), C.c_long(rows), C.c_long(cols),
                                  C.c_void_p(want.data_ptr()), st) == 0
    assert L.lfamd_launch_pk_expand(t, C.c_void_p(W.data.data_ptr()), C.c_long(rows), C.c_long(cols), C.c_void_p(got.data_ptr()), st) == 0
    torch.cuda.synchronize()
    assert torch.equal(got, want)


@pytest.mark.parametrize("shape", [(64, 512), (37, 1024), (7, 256)], ids=str)
def test_compact_iq4_xs_image_expands_to_the_byte_image(gpu, shape):
    """IQ4_XS is resident as codebook INDICES on the nibble lattice (144 bytes per 256 weights); batches expand it per call into the
    byte image (codebook value + 128) the MFMA body reads: pack + expand must be bit-for-bit the byte image built from the GGUF rows."""
    import ctypes as C
    import torch
    from llamafile_amd import _hip
    rows, cols = shape
    t = T.IQ4_XS
    raw = synth.random_weights(t, rows, cols, seed=19)
    W = gpu.upload_weights(t, raw, rows, cols)
    L = C.CDLL(_hip.HIP_SO)
    L.lfamd_wprep8_bytes.restype = C.c_size_t
    nbytes = L.lfamd_wprep8_bytes(C.c_long(rows), C.c_long(cols))
    assert W.data.numel() < 0.6 * nbytes
    rawd = torch.from_numpy(raw).cuda()
    want = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    got = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.lfamd_launch_wprep8(t, C.c_void_p(rawd.data_ptr()), C.c_size_t(raw.shape[1]), C.c_long(rows), C.c_long(cols),
                                 C.c_void_p(want.data_ptr()), st) == 0
    assert L.lfamd_launch_pk4x_expand(C.c_void_p(W.data.data_ptr()), C.c_long(rows), C.c_long(cols), C.c_void_p(got.data_ptr()), st) == 0
    torch.cuda.synchronize()
    assert torch.equal(got, want)
