"""GGUF files for the mat-mul path: a ctypes binding of the host library's reader (llamafile_amd/csrc/gguf_reader.cpp,
include/llamafile_sgemm.h) and a small WRITER used to build synthetic model files for tests and for running BASELINE's
configs end to end without real checkpoints (public GGUF v3 layout)."""
from __future__ import annotations

import ctypes as C
import struct

import numpy as np

from . import _hip, ggml_types as T

_lib = None


def _host():
    global _lib
    if _lib is None:
        L = C.CDLL(_hip.HOST_SO)
        L.lfamd_gguf_open.restype = C.c_void_p
        L.lfamd_gguf_open.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.lfamd_gguf_close.argtypes = [C.c_void_p]
        for n in ("lfamd_gguf_n_tensors", "lfamd_gguf_n_kv", "lfamd_gguf_find_tensor"):
            getattr(L, n).restype = C.c_long
        L.lfamd_gguf_n_tensors.argtypes = [C.c_void_p]
        L.lfamd_gguf_n_kv.argtypes = [C.c_void_p]
        L.lfamd_gguf_version.argtypes = [C.c_void_p]
        L.lfamd_gguf_alignment.argtypes = [C.c_void_p]
        L.lfamd_gguf_alignment.restype = C.c_size_t
        L.lfamd_gguf_find_tensor.argtypes = [C.c_void_p, C.c_char_p]
        L.lfamd_gguf_tensor.argtypes = [C.c_void_p, C.c_long, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                        C.POINTER(C.c_int64 * 4), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.lfamd_gguf_get_u64.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64)]
        L.lfamd_gguf_get_f64.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)]
        L.lfamd_gguf_get_str.argtypes = [C.c_void_p, C.c_char_p]
        L.lfamd_gguf_get_str.restype = C.c_char_p
        _lib = L
    return _lib


class GGUFTensor:
    __slots__ = ("name", "type", "ne", "ptr", "nbytes")

    def array(self) -> np.ndarray:
        """uint8 view [rows, row_bytes] of the mapped bytes (read-only)."""
        rows = int(self.ne[1] * self.ne[2] * self.ne[3])
        buf = (C.c_uint8 * self.nbytes).from_address(self.ptr)
        a = np.frombuffer(buf, dtype=np.uint8)
        a.flags.writeable = False
        return a.reshape(rows, self.nbytes // max(rows, 1))


class GGUFFile:
    def __init__(self, path: str):
        err = C.create_string_buffer(256)
        self.L = _host()
        self.h = self.L.lfamd_gguf_open(str(path).encode(), err, 256)
        if not self.h:
            raise ValueError(f"{path}: {err.value.decode()}")
        self.version = self.L.lfamd_gguf_version(self.h)
        self.alignment = self.L.lfamd_gguf_alignment(self.h)
        self.tensors = []
        for i in range(self.L.lfamd_gguf_n_tensors(self.h)):
            name, typ, nd, ne, data, nb = C.c_char_p(), C.c_int(), C.c_int(), (C.c_int64 * 4)(), C.c_void_p(), C.c_size_t()
            assert self.L.lfamd_gguf_tensor(self.h, i, C.byref(name), C.byref(typ), C.byref(nd), C.byref(ne), C.byref(data), C.byref(nb)) == 0
            t = GGUFTensor()
            t.name, t.type, t.ne, t.ptr, t.nbytes = name.value.decode(), typ.value, tuple(ne), data.value, nb.value
            self.tensors.append(t)

    def tensor(self, name: str) -> GGUFTensor:
        i = self.L.lfamd_gguf_find_tensor(self.h, name.encode())
        if i < 0:
            raise KeyError(name)
        return self.tensors[i]

    def get(self, key: str):
        u, f = C.c_uint64(), C.c_double()
        if self.L.lfamd_gguf_get_u64(self.h, key.encode(), C.byref(u)) == 0:
            return u.value
        if self.L.lfamd_gguf_get_f64(self.h, key.encode(), C.byref(f)) == 0:
            return f.value
        s = self.L.lfamd_gguf_get_str(self.h, key.encode())
        return s.decode() if s is not None else None

    def close(self):
        if self.h:
            self.L.lfamd_gguf_close(self.h)
            self.h = None


def _s(b: str) -> bytes:
    e = b.encode()
    return struct.pack("<Q", len(e)) + e


def write_gguf(path, metadata: dict, tensors: list, alignment: int = 32, version: int = 3) -> None:
    """tensors: [(name, ggml_type, (ne0, ne1[, ne2]), uint8 array of the raw blocks)].  metadata values: int (u32 / u64),
    float (f32), str, bool, or a list of str / int (array)."""
    kv = b""
    meta = dict(metadata)
    if alignment != 32:
        meta["general.alignment"] = alignment
    for k, v in meta.items():
        kv += _s(k)
        if isinstance(v, bool):
            kv += struct.pack("<IB", 7, int(v))
        elif isinstance(v, int):
            kv += struct.pack("<II", 4, v) if v < 2 ** 32 else struct.pack("<IQ", 10, v)
        elif isinstance(v, float):
            kv += struct.pack("<If", 6, v)
        elif isinstance(v, str):
            kv += struct.pack("<I", 8) + _s(v)
        elif isinstance(v, list) and v and isinstance(v[0], str):
            kv += struct.pack("<IIQ", 9, 8, len(v)) + b"".join(_s(x) for x in v)
        elif isinstance(v, list):
            kv += struct.pack("<IIQ", 9, 5, len(v)) + struct.pack(f"<{len(v)}i", *v)
        else:
            raise TypeError(k)
    infos, blobs, off = b"", [], 0
    for name, typ, ne, raw in tensors:
        raw = np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
        infos += _s(name) + struct.pack("<I", len(ne)) + struct.pack(f"<{len(ne)}Q", *ne) + struct.pack("<IQ", typ, off)
        blobs.append((off, raw))
        off = (off + raw.size + alignment - 1) // alignment * alignment
    head = b"GGUF" + struct.pack("<IQQ", version, len(tensors), len(meta)) + kv + infos
    pad = (-len(head)) % alignment
    with open(path, "wb") as f:
        f.write(head + b"\0" * pad)
        base = f.tell()
        for o, raw in blobs:
            f.seek(base + o)
            f.write(raw.tobytes())
        f.truncate(base + off if off else base)
