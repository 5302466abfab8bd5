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


def _tiny_llama(tmp_path, n_layers=2, n_embd=512, n_ff=1024, n_kv=256, n_vocab=768):
    ts = []
    seed = 100
    for il in range(n_layers):
        hi = T.Q6_K if il == 0 else T.Q4_K
        for name, t, m, k in (("attn_q", T.Q4_K, n_embd, n_embd), ("attn_k", T.Q4_K, n_kv, n_embd), ("attn_v", hi, n_kv, n_embd),
                              ("attn_output", T.Q4_K, n_embd, n_embd), ("ffn_gate", T.Q4_K, n_ff, n_embd),
                              ("ffn_up", T.Q4_K, n_ff, n_embd), ("ffn_down", hi, n_embd, n_ff)):
            seed += 1
            ts.append((f"blk.{il}.{name}.weight", t, (k, m), synth.random_weights(t, m, k, seed)))
        ts.append((f"blk.{il}.attn_norm.weight", T.F32, (n_embd,), synth.random_weights(T.F32, 1, n_embd, seed + 50)))
    ts.append(("output.weight", T.Q6_K, (n_embd, n_vocab), synth.random_weights(T.Q6_K, n_vocab, n_embd, 999)))
    p = tmp_path / "tiny.gguf"
    gguf.write_gguf(p, {"general.architecture": "llama", "llama.block_count": n_layers}, ts)
    return p, ts


def test_op_list_from_a_model_file(tmp_path):
    """bench.py --gguf: the mat-mul inventory comes from the file's tensor directory (llama-bench / localscore run from a
    model file, localscore/benchmark.cpp:93-145) — names, stored types and shapes, grouping inputs, shard modes."""
    from llamafile_amd import llama_shapes as LS
    p, ts = _tiny_llama(tmp_path)
    g = gguf.GGUFFile(p)
    layers, tensors = LS.from_gguf(g)
    assert len(layers) == 3 and [len(l) for l in layers] == [7, 7, 1]
    assert layers[0][2] == LS.MatMul("blk.0.attn_v", T.Q6_K, 256, 512, "attn_in", "rows")
    assert layers[1][6] == LS.MatMul("blk.1.ffn_down", T.Q4_K, 512, 1024, "ffn_down_in", "cols")
    assert layers[2][0] == LS.MatMul("output", T.Q6_K, 768, 512, "out_in", "vocab")
    for ops in layers:
        for o in ops:
            raw = next(r for n, _, _, r in ts if n == o.name + ".weight")
            assert np.array_equal(tensors[o.name].array().reshape(-1), np.ascontiguousarray(raw).view(np.uint8).reshape(-1))
    g.close()


@pytest.mark.gpu
def test_bench_runs_from_a_model_file(gpu, tmp_path):
    """The whole harness on a file: `bench.py --gguf tiny.gguf` prints the contract's JSON line with the file named in it."""
    import json
    import os
    import subprocess
    import sys
    p, _ = _tiny_llama(tmp_path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gguf", str(p), "--steps", "1", "--warmup", "1", "--prefill", "128",
                        "--decode", "4", "--no-cpu-baseline", "--no-extra-configs"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["value"] > 0 and d["config"]["model"] == "gguf:tiny.gguf" and "tiny.gguf" in d["data"]
    assert d["roofline"]["frac"] > 0 and d["config"]["decode_launches_per_pass"] == 2 * 4 + 1
