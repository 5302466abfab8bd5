"""PCIe-inclusive HOST-POINTER path (the drop-in boundary of llamafile/sgemm.h): llamafile_sgemm called with host A, B, C
exactly like ggml's CPU executor does, 225 calls per decoded token (7 per layer x 32 + output.weight), weights registered
as immutable (kept packed on the device), activations pre-quantised to Q8_K on the host like ggml_compute_forward_mul_mat
does.  Never part of bench.py's `value`; prints one JSON object for profiles/."""
import ctypes as C
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from llamafile_amd import _hip, ggml_types as T, llama_shapes as LS, synth  # noqa: E402

host = C.CDLL(_hip.HOST_SO)
host.llamafile_sgemm.restype = C.c_bool
host.llamafile_sgemm.argtypes = [C.c_long] * 3 + [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_long] + [C.c_int] * 5
host.llamafile_sgemm_amd_register_weights.argtypes = [C.c_void_p, C.c_size_t]
assert host.llamafile_sgemm_amd_available() == 1

layers = LS.llama3_8b_q4_k_m()
n_layers = int(sys.argv[1]) if len(sys.argv) > 1 else 4  # distinct layers held on the host (all 32 = 4.6 GB)
ops = []
for li in list(range(n_layers)) + [len(layers) - 1]:
    for s in layers[li]:
        A = synth.random_weights(s.type, s.m, s.k, 100 + len(ops))
        host.llamafile_sgemm_amd_register_weights(A.ctypes.data, A.nbytes)
        ops.append((s, A))
res = {}
for n in (1, 512):
    calls = []
    for s, A in ops:
        B = synth.quantize_activations(T.Q8_K, synth.random_activations(n, s.k, 7))
        Cm = np.empty((n, s.m), dtype=np.float32)
        calls.append((s, A, B, Cm))

    def one_pass():
        for s, A, B, Cm in calls:
            kb = s.k // 256
            assert host.llamafile_sgemm(s.m, n, kb, A.ctypes.data, kb, B.ctypes.data, kb, Cm.ctypes.data, s.m, 0, 1, s.type, T.Q8_K, T.F32)

    one_pass()  # uploads + packs the registered weights
    reps = 5 if n == 1 else 2
    t0 = time.perf_counter()
    for _ in range(reps):
        one_pass()
    dt = (time.perf_counter() - t0) / reps
    per_layer_calls = len(calls) - 1
    layer_time = dt * per_layer_calls / len(calls)  # (approximation: calls weighted equally for the scale-up below)
    full = dt / len(calls) * 225
    res[f"n={n}"] = {"calls_timed": len(calls), "us_per_call": round(dt / len(calls) * 1e6, 1),
                     "ms_per_225_call_pass_extrapolated": round(full * 1e3, 2),
                     "tokens_per_s": round(n / full, 1)}
print(json.dumps({"what": "llamafile_sgemm with HOST pointers (PCIe-inclusive): B uploaded, C downloaded, stream synchronised per call; "
                          "weights registered immutable and kept packed on the device", "layers_held": n_layers, **res}))
