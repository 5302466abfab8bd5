"""Numpy statement of the packed weight layouts (llamafile_amd/csrc/lfamd_device.h), written
independently of the device pack kernels so the two can be compared byte for byte."""
import numpy as np


def nibpos(j):
    return (j >> 1) + 4 * (j & 1)


def qhbit(dd, j):
    f = (2, 4, 6, 0)
    return 16 * (j & 1) + 2 * (f[j >> 1] + (dd & 1))


def q4k_codes(raw_rows, nb):
    """raw [rows, nb*144] -> codes [rows, nb, 256] (0..15) and headers [rows, nb, 16]."""
    rows = raw_rows.shape[0]
    blk = raw_rows.reshape(rows, nb, 144)
    qs = blk[:, :, 16:].reshape(rows, nb, 4, 32)
    codes = np.empty((rows, nb, 4, 64), dtype=np.uint8)
    codes[..., :32] = qs & 15
    codes[..., 32:] = qs >> 4
    return codes.reshape(rows, nb, 256), blk[:, :, :16]


def q6k_codes(raw_rows, nb):
    rows = raw_rows.shape[0]
    blk = raw_rows.reshape(rows, nb, 210)
    ql = blk[:, :, :128].reshape(rows, nb, 2, 64)
    qh = blk[:, :, 128:192].reshape(rows, nb, 2, 32)
    codes = np.empty((rows, nb, 2, 128), dtype=np.uint8)
    codes[..., 0:32] = (ql[..., 0:32] & 15) | (((qh >> 0) & 3) << 4)
    codes[..., 32:64] = (ql[..., 32:64] & 15) | (((qh >> 2) & 3) << 4)
    codes[..., 64:96] = (ql[..., 0:32] >> 4) | (((qh >> 4) & 3) << 4)
    codes[..., 96:128] = (ql[..., 32:64] >> 4) | (((qh >> 6) & 3) << 4)
    return codes.reshape(rows, nb, 256), blk[:, :, 192:208], blk[:, :, 208:210]


def _pad_rows(a, mult):
    rows = a.shape[0]
    pad = (-rows) % mult
    if pad:
        a = np.concatenate([a, np.zeros((pad,) + a.shape[1:], dtype=a.dtype)], axis=0)
    return a


def _nibble_plane(codes4, nrt, nb):
    """codes4 [nrt*32, nb, 256] 4-bit values -> dwords [nrt, nb, 4(g), 64(lane), 4(dd)]."""
    c = codes4.reshape(nrt, 32, nb, 4, 4, 2, 8).astype(np.uint32)  # [rt, i, b, g, dd, h, j]
    out = np.zeros((nrt, nb, 4, 2, 32, 4), dtype=np.uint32)  # [rt, b, g, h, i, dd]
    for j in range(8):
        out |= c[..., j].transpose(0, 2, 3, 5, 1, 4) << np.uint32(4 * nibpos(j))
    return out.reshape(nrt, nb, 4, 64, 4)


def pack_q4k(raw, rows, cols):
    nb = cols // 256
    codes, hdr = q4k_codes(raw[:, : nb * 144], nb)
    codes, hdr = _pad_rows(codes, 32), _pad_rows(hdr, 32)
    nrt = codes.shape[0] // 32
    qs = _nibble_plane(codes, nrt, nb)
    out = np.zeros((nrt, nb, 4608), dtype=np.uint8)
    out[:, :, :4096] = qs.view(np.uint8).reshape(nrt, nb, 4096)
    out[:, :, 4096:] = hdr.reshape(nrt, 32, nb, 16).transpose(0, 2, 1, 3).reshape(nrt, nb, 512)
    return out.reshape(-1)


def pack_q6k(raw, rows, cols):
    nb = cols // 256
    codes, sc, d = q6k_codes(raw[:, : nb * 210], nb)
    codes, sc, d = _pad_rows(codes, 32), _pad_rows(sc, 32), _pad_rows(d, 32)
    nrt = codes.shape[0] // 32
    ql = _nibble_plane(codes & 15, nrt, nb)
    hi = (codes >> 4).reshape(nrt, 32, nb, 4, 4, 2, 8).astype(np.uint32)  # [rt, i, b, g, dd, h, j]
    qh = np.zeros((nrt, nb, 2, 2, 32, 2, 2), dtype=np.uint32)  # [rt, b, gg, h, i, g&1, e]
    for dd in range(4):
        for j in range(8):
            v = hi[:, :, :, :, dd, :, j]  # [rt, i, b, g, h]
            v = v.reshape(nrt, 32, nb, 2, 2, 2)  # g -> (gg, g&1)
            qh[..., dd >> 1] |= v.transpose(0, 2, 3, 5, 1, 4) << np.uint32(qhbit(dd, j))
    out = np.zeros((nrt, nb, 6720), dtype=np.uint8)
    out[:, :, :4096] = ql.view(np.uint8).reshape(nrt, nb, 4096)
    out[:, :, 4096:6144] = qh.reshape(nrt, nb, 2, 64, 4).view(np.uint8).reshape(nrt, nb, 2048)
    out[:, :, 6144:6656] = sc.reshape(nrt, 32, nb, 16).transpose(0, 2, 1, 3).reshape(nrt, nb, 512)
    out[:, :, 6656:] = d.reshape(nrt, 32, nb, 2).transpose(0, 2, 1, 3).reshape(nrt, nb, 64)
    return out.reshape(-1)


def pack_q80(raw, rows, cols):
    nblk = cols // 32
    blk = raw[:, : nblk * 34].reshape(rows, nblk, 34)
    padb = (-nblk) % 4
    if padb:
        blk = np.concatenate([blk, np.zeros((rows, padb, 34), dtype=np.uint8)], axis=1)
    blk = _pad_rows(blk, 8)
    nrg, nq = blk.shape[0] // 8, blk.shape[1] // 4
    b = blk.reshape(nrg, 8, nq, 4, 34)  # [rg, r, L, dd, byte]
    qs = b[..., 2:].reshape(nrg, 8, nq, 4, 8, 4)  # [rg, r, L, dd, j, byte]
    out = np.zeros((nrg, nq, 1088), dtype=np.uint8)
    out[:, :, :1024] = qs.transpose(0, 2, 1, 4, 3, 5).reshape(nrg, nq, 1024)  # [rg, L, r, j, dd, byte]
    out[:, :, 1024:] = b[..., :2].transpose(0, 2, 1, 3, 4).reshape(nrg, nq, 64)  # [rg, L, r, dd, 2]
    return out.reshape(-1)
