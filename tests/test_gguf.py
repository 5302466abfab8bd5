"""GGUF reader of the host library (f-2): directory, metadata, alignment, error cases on synthetic files; and (GPU) a
mat-mul straight from the mapped file through llamafile_sgemm, whose weight cache keeps read-only mappings."""
import ctypes as C
import struct

import numpy as np
import pytest

from llamafile_amd import _hip, ggml_types as T, gguf, synth
from helpers import rel_err


def _model(tmp_path, alignment=32):
    k, m = 512, 96
    ts = [("blk.0.attn_q.weight", T.Q4_K, (k, m), synth.random_weights(T.Q4_K, m, k, 1)),
          ("blk.0.attn_v.weight", T.Q6_K, (k, 32), synth.random_weights(T.Q6_K, 32, k, 2)),
          ("blk.0.ffn_gate_exps.weight", T.Q8_0, (256, 16, 4), np.concatenate([synth.random_weights(T.Q8_0, 16, 256, 3 + e) for e in range(4)])),
          ("output_norm.weight", T.F32, (k,), synth.random_weights(T.F32, 1, k, 9))]
    p = tmp_path / "m.gguf"
    gguf.write_gguf(p, {"general.architecture": "llama", "llama.block_count": 1, "llama.rope.freq_base": 500000.0,
                        "llama.big": 2 ** 40, "general.ok": True, "tokenizer.ggml.tokens": ["a", "bc", "def"],
                        "tokenizer.ggml.token_type": [1, 2, 3]}, ts, alignment=alignment)
    return p, ts


@pytest.mark.parametrize("alignment", [32, 64, 4096])
def test_directory_metadata_and_bytes(tmp_path, alignment):
    p, ts = _model(tmp_path, alignment)
    g = gguf.GGUFFile(p)
    assert g.version == 3 and g.alignment == alignment and len(g.tensors) == len(ts)
    assert g.get("general.architecture") == "llama" and g.get("llama.block_count") == 1 and g.get("llama.big") == 2 ** 40
    assert abs(g.get("llama.rope.freq_base") - 500000.0) < 1e-3 and g.get("general.ok") == 1 and g.get("missing") is None
    for (name, typ, ne, raw), t in zip(ts, g.tensors):
        assert t.name == name and t.type == typ and t.ne[:len(ne)] == tuple(ne) and all(d == 1 for d in t.ne[len(ne):])
        assert t.ptr % alignment == 0 or alignment > 4096
        assert np.array_equal(t.array().reshape(-1), np.ascontiguousarray(raw).view(np.uint8).reshape(-1))
    assert g.tensor("blk.0.attn_v.weight").type == T.Q6_K
    with pytest.raises(KeyError):
        g.tensor("nope")
    g.close()


def test_rejects_malformed_files(tmp_path):
    p, _ = _model(tmp_path)
    good = p.read_bytes()
    cases = {"magic": b"GGUX" + good[4:], "v1": good[:4] + struct.pack("<I", 1) + good[8:], "truncated": good[:200],
             "short": good[:10], "data cut": good[:-100]}
    for name, blob in cases.items():
        q = tmp_path / f"bad_{name.replace(' ', '_')}.gguf"
        q.write_bytes(blob)
        with pytest.raises(ValueError):
            gguf.GGUFFile(q)
    # a row length that is not a multiple of the block size (gguf_init_from_file's check)
    q = tmp_path / "bad_blk.gguf"
    gguf.write_gguf(q, {}, [("w", T.Q4_K, (100, 4), np.zeros(4 * 144, dtype=np.uint8))])
    with pytest.raises(ValueError):
        gguf.GGUFFile(q)


def test_rejects_wrapping_offsets_and_sizes(tmp_path):
    """A crafted tensor directory must not wrap the bounds arithmetic: an aligned offset near 2^64 (offset + nbytes wraps to a
    small sum) and dimensions whose byte count overflows are both refused at open."""
    q = tmp_path / "one.gguf"
    gguf.write_gguf(q, {}, [("w", T.F32, (64, 4), np.zeros(64 * 4 * 4, dtype=np.uint8))])
    good = bytearray(q.read_bytes())
    # tensor info = name (u64 len + bytes), n_dims u32, ne[2] u64, type u32, offset u64 — the last 8 + 4 + 16 bytes before the
    # aligned data section
    info_end = good.index(b"w", 24) + 1 + 4 + 16 + 4 + 8
    off_pos, ne_pos = info_end - 8, info_end - 8 - 4 - 16
    assert struct.unpack_from("<Q", good, off_pos)[0] == 0 and struct.unpack_from("<QQ", good, ne_pos) == (64, 4)
    gguf.GGUFFile(q).close()
    for name, pos, val in [("offset near 2^64", off_pos, struct.pack("<Q", 2 ** 64 - 32)),
                           ("offset past the end", off_pos, struct.pack("<Q", 1 << 40)),
                           ("dims overflow bytes", ne_pos, struct.pack("<QQ", 2 ** 62, 1)),
                           ("dims overflow product", ne_pos, struct.pack("<QQ", 2 ** 40, 2 ** 40))]:
        bad = bytearray(good)
        bad[pos:pos + len(val)] = val
        b = tmp_path / ("bad_" + name.replace(" ", "_").replace("^", "") + ".gguf")
        b.write_bytes(bytes(bad))
        with pytest.raises(ValueError):
            gguf.GGUFFile(b)


@pytest.mark.gpu
def test_sgemm_straight_from_the_mapped_file(gpu, oracle, tmp_path):
    p, ts = _model(tmp_path)
    g = gguf.GGUFFile(p)
    host = C.CDLL(_hip.HOST_SO)
    host.llamafile_sgemm.restype = C.c_bool
    host.llamafile_sgemm.argtypes = [C.c_long] * 3 + [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_long] + [C.c_int] * 5
    host.llamafile_sgemm_amd_cached_bytes.restype = C.c_size_t
    before = host.llamafile_sgemm_amd_cached_bytes()
    for name in ("blk.0.attn_q.weight", "blk.0.attn_v.weight"):
        t = g.tensor(name)
        k, m = t.ne[0], t.ne[1]
        x = synth.random_activations(3, k, 5)
        B = synth.quantize_activations(T.Q8_K, x)
        out = np.zeros((3, m), dtype=np.float32)
        for _ in range(2):
            assert host.llamafile_sgemm(m, 3, k // 256, t.ptr, k // 256, B.ctypes.data, k // 256, out.ctypes.data, m, 0, 1, t.type, T.Q8_K, T.F32)
        ok, G = oracle.sgemm(t.type, t.array(), T.Q8_K, B, m, 3, k)
        assert ok == 1 and rel_err(out, G) <= 2e-6
    assert host.llamafile_sgemm_amd_cached_bytes() > before  # the read-only mapping was recognised: packed once, kept
    host.llamafile_sgemm_amd_reset()
    g.close()
