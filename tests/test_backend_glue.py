"""The GPU-module boundary: the 12 ms_abi symbols llamafile/cuda.c:726-737 imports, driven
from a C host program that plays llamafile's side (tests/backend_host/backend_host.c): dlopen, symbol import,
ggml_cuda_link with a ggml_backend_api callback table, buffers, set_tensor, supports_op, graph_compute, get_tensor."""
import os
import subprocess

import numpy as np
import pytest

from llamafile_amd import _hip, ggml_types as T, synth
from helpers import rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "backend_host", "backend_host.c")
EXE = os.path.join(ROOT, "tests", "backend_host", "backend_host")


@pytest.fixture(scope="module")
def host_exe():
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(SRC):
        subprocess.check_call(["gcc", "-O1", "-Wall", "-o", EXE, SRC, "-ldl"])
    return EXE


def test_module_exports_the_twelve_symbols_and_links_or_declines(host_exe):
    """On a box without an MI355X ggml_cuda_link answers false (llamafile then falls back to the CPU, cuda.c:744-752)."""
    out = subprocess.run([host_exe, _hip.HIP_SO, "exports"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split()[0] in ("linked", "nolink")


@pytest.mark.gpu
def test_a_host_with_other_struct_layouts_is_declined(gpu, host_exe, tmp_path):
    """The recalled struct layouts are cross-checked through the host's own accessors (ggml_nbytes, ggml_nelements,
    ggml_element_size, ggml_is_contiguous, ggml_backend_buffer_get_usage): when they disagree with what the module reads from the
    structs, supports_op says no (the host program exits 11 = "supports_op says no") and the module says why, once."""
    t, m, k, n = T.Q4_K, 64, 512, 3
    wp, xp, op = tmp_path / "w.bin", tmp_path / "x.bin", tmp_path / "o.bin"
    synth.random_weights(t, m, k, 7).tofile(wp)
    synth.random_activations(n, k, 8).tofile(xp)
    args = [host_exe, _hip.HIP_SO, "mulmat", str(t), str(m), str(k), str(n), "1", str(wp), str(xp), str(op)]
    ok = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0, ok.stderr
    bad = subprocess.run(args, capture_output=True, text=True, timeout=300, env={**os.environ, "BACKEND_HOST_SKEW": "1"})
    assert bad.returncode == 11, (bad.returncode, bad.stderr)
    assert "do not match this module's layouts" in bad.stderr


@pytest.mark.gpu
def test_matrices_only_switch_keeps_layer_buffers_on_the_host(gpu, host_exe):
    """with LFAMD_BACKEND_MATRICES_ONLY=1 only the split (matrix) buffer type is device memory, so norm
    weights, the KV cache and compute buffers stay with ggml's CPU backend."""
    for env, want in (({}, "layer=device matrix=device"), ({"LFAMD_BACKEND_MATRICES_ONLY": "1"}, "layer=host matrix=device")):
        r = subprocess.run([host_exe, _hip.HIP_SO, "bufts"], capture_output=True, text=True, timeout=120, env={**os.environ, **env})
        assert r.returncode == 0 and r.stdout.strip() == want, (r.stdout, r.stderr)


def check_mul_mat(oracle, got, t, W, x, m, k, n, nb2):
    bt = T.VEC_DOT[t]
    if t == T.F16:
        # ggml hands f16 weights their activations rounded to f16 (n > 2) or as f32 (n <= 2, tinyblas_cpu_sgemm.inc:121-136);
        # the module keeps f32 activations for n <= 8: at least as accurate — within f16 rounding of either form
        G16 = oracle.f64_gemm(t, W, T.F16, synth.quantize_activations(T.F16, x), m, n * nb2, k)
        G32 = oracle.f64_gemm(t, W, T.F32, synth.quantize_activations(T.F32, x), m, n * nb2, k)
        assert min(rel_err(got, G16), rel_err(got, G32)) <= 2e-6 and rel_err(got, G16) <= 1e-3
        return
    v = oracle.variant("zen4")  # the module passes LFAMD_FLAG_Q0_VREGS32 (AVX512 build of tinyBLAS_Q0)
    for s in range(nb2):  # every slice is its own llamafile_sgemm problem (the Q0 Kahan geometry depends on n)
        ok, G = oracle.sgemm(t, W, bt, synth.quantize_activations(bt, x[s * n:(s + 1) * n]), m, n, k, v=v)
        assert ok == 1
        g = got[s * n:(s + 1) * n]
        if t == T.Q8_0:
            assert np.array_equal(g.view(np.uint32), G.view(np.uint32))
        else:
            assert rel_err(g, G) <= (1e-3 if n > 8 else 2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("t,m,k,n,nb2", [(T.Q4_K, 96, 1024, 1, 1), (T.Q4_K, 160, 768, 40, 1), (T.Q6_K, 64, 512, 3, 2), (T.Q8_0, 72, 256, 1, 1),
                                         (T.F16, 48, 256, 5, 3), (T.Q5_K, 32, 512, 12, 1), (T.Q4_0, 64, 256, 2, 1)],
                         ids=lambda v: str(v))
def test_mul_mat_node_through_the_backend_interface(gpu, oracle, host_exe, tmp_path, t, m, k, n, nb2):
    """GGML_OP_MUL_MAT with f32 src1 and dims-2 broadcast (src1 has nb2 slices, src0 one): what
    ggml_compute_forward_mul_mat computes — quantise src1 rows to the type's vec_dot format, llamafile_sgemm per slice."""
    W = synth.random_weights(t, m, k, 7)
    x = synth.random_activations(n * nb2, k, 8)
    wp, xp, op = tmp_path / "w.bin", tmp_path / "x.bin", tmp_path / "o.bin"
    W.tofile(wp)
    x.tofile(xp)
    r = subprocess.run([host_exe, _hip.HIP_SO, "mulmat", str(t), str(m), str(k), str(n), str(nb2), str(wp), str(xp), str(op)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr
    got = np.fromfile(op, dtype=np.float32).reshape(nb2 * n, m)
    check_mul_mat(oracle, got, t, W, x, m, k, n, nb2)


@pytest.mark.gpu
@pytest.mark.parametrize("tokens,tasks", [(1, 1), (3, 2), (20, 1)])
def test_mul_mat_id_node_through_the_backend_interface(gpu, oracle, host_exe, tmp_path, tokens, tasks):
    t, m, k, experts, thinkers = T.Q4_K, 64, 512, 6, 2
    W = np.stack([synth.random_weights(t, m, k, 50 + e) for e in range(experts)])
    x = synth.random_activations(tokens * tasks, k, 9).reshape(tokens, tasks, k)
    rng = np.random.default_rng(4)
    ids = np.stack([rng.permutation(experts)[:thinkers] for _ in range(tokens)]).astype(np.int32)
    wp, xp, ip, op = (tmp_path / n for n in ("w.bin", "x.bin", "i.bin", "o.bin"))
    W.tofile(wp), x.tofile(xp), ids.tofile(ip)
    r = subprocess.run([host_exe, _hip.HIP_SO, "mulmatid", str(t), str(m), str(k), str(experts), str(thinkers), str(tasks), str(tokens),
                        str(wp), str(xp), str(ip), str(op)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr
    got = np.fromfile(op, dtype=np.float32).reshape(tokens, thinkers, m)
    q = oracle.quantize(T.Q8_K, x.reshape(-1, k))
    G = np.zeros_like(got)
    for tk in range(tokens):
        for th in range(thinkers):
            ok, c = oracle.sgemm(t, W[ids[tk, th]], T.Q8_K, q[tk * tasks + th % tasks][None, :], m, 1, k)
            assert ok == 1
            G[tk, th] = c[0]
    # (normwise over the whole result like the other batch tests: more than 4 tokens run the grouped MFMA launch on scaled
    # f16 operands, whose per-output error is ~7e-4 of the outputs' rms at k = 512)
    assert rel_err(got, G) <= (1e-3 if tokens > 4 else 2e-6)


TWO = {"LFAMD_BACKEND_DEVICES": "0,0"}  # two logical devices on the one GPU of the box: the multi-device paths, rehearsed


@pytest.mark.gpu
def test_device_list_is_what_the_module_reports(gpu, host_exe):
    """ggml_backend_cuda_get_device_count (ggml-cuda.cu.patch:19532): every gfx950 device of the process by default,
    LFAMD_BACKEND_DEVICES restricts / repeats; ordinals that do not exist are dropped."""
    import torch
    visible = torch.cuda.device_count()
    for env, want in (({}, f"count={visible}"), (TWO, "count=2"), ({"LFAMD_BACKEND_DEVICES": "0,99,0,0"}, "count=3"),
                      ({"LFAMD_BACKEND_DEVICES": "99"}, "count=0")):
        r = subprocess.run([host_exe, _hip.HIP_SO, "exports"], capture_output=True, text=True, timeout=120, env={**os.environ, **env})
        assert r.returncode == 0 and r.stdout.split()[1] == want, (env, r.stdout, r.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("t,m,k,n,nb2,split,main", [(T.Q4_K, 1024, 512, 1, 1, "1,1", 0), (T.Q4_K, 1280, 512, 40, 1, "0.3,0.7", 1),
                                                    (T.Q6_K, 768, 512, 3, 2, "1,1", 0), (T.Q8_0, 512, 256, 1, 1, "0,0", 1),
                                                    (T.F16, 520, 256, 5, 1, "1,3", 0), (T.Q4_K, 96, 1024, 2, 1, "1,1", 0),
                                                    (T.Q5_K, 512, 512, 12, 1, "1,0", 1)], ids=lambda v: str(v))
def test_row_split_mul_mat_over_two_devices(gpu, oracle, host_exe, tmp_path, t, m, k, n, nb2, split, main):
    """--split-mode row (ggml_backend_cuda_split_buffer_type, ggml-cuda.cu.patch:17123-17450; ggml_cuda_op_mul_mat with split
    src0, :18060-18330): the matrix's rows are cut at the tensor_split fractions over the logical devices, each computes its
    columns of the result, dst is assembled on the main device.  Same oracle, same tolerances as the one-device node; the host
    program also checks set_tensor / get_tensor round-trip, clear() and the re-upload on the split buffer."""
    W = synth.random_weights(t, m, k, 7)
    x = synth.random_activations(n * nb2, k, 8)
    wp, xp, op = tmp_path / "w.bin", tmp_path / "x.bin", tmp_path / "o.bin"
    W.tofile(wp)
    x.tofile(xp)
    r = subprocess.run([host_exe, _hip.HIP_SO, "mulmat", str(t), str(m), str(k), str(n), str(nb2), str(wp), str(xp), str(op)],
                       capture_output=True, text=True, timeout=300,
                       env={**os.environ, **TWO, "BACKEND_HOST_SPLIT": split, "BACKEND_HOST_MAIN": str(main)})
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr
    assert "weights in ROCm-lfamd_Split" in r.stderr
    check_mul_mat(oracle, np.fromfile(op, dtype=np.float32).reshape(nb2 * n, m), t, W, x, m, k, n, nb2)


@pytest.mark.gpu
def test_expert_stack_in_a_split_buffer_stays_whole(gpu, oracle, host_exe, tmp_path):
    """llama.cpp places expert tensors in the matrix buffer type too; the reference refuses MUL_MAT_ID there
    (ggml-cuda.cu.patch:18501), this module keeps the stack whole on the first device and serves it."""
    t, m, k, experts, thinkers, tokens, tasks = T.Q4_K, 64, 512, 6, 2, 3, 1
    W = np.stack([synth.random_weights(t, m, k, 50 + e) for e in range(experts)])
    x = synth.random_activations(tokens * tasks, k, 9).reshape(tokens, tasks, k)
    rng = np.random.default_rng(4)
    ids = np.stack([rng.permutation(experts)[:thinkers] for _ in range(tokens)]).astype(np.int32)
    wp, xp, ip, op = (tmp_path / n for n in ("w.bin", "x.bin", "i.bin", "o.bin"))
    W.tofile(wp), x.tofile(xp), ids.tofile(ip)
    r = subprocess.run([host_exe, _hip.HIP_SO, "mulmatid", str(t), str(m), str(k), str(experts), str(thinkers), str(tasks), str(tokens),
                        str(wp), str(xp), str(ip), str(op)], capture_output=True, text=True, timeout=300,
                       env={**os.environ, **TWO, "BACKEND_HOST_SPLIT": "1,1"})
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr
    got = np.fromfile(op, dtype=np.float32).reshape(tokens, thinkers, m)
    q = oracle.quantize(T.Q8_K, x.reshape(-1, k))
    for tk in range(tokens):
        for th in range(thinkers):
            ok, c = oracle.sgemm(t, W[ids[tk, th]], T.Q8_K, q[tk * tasks + th % tasks][None, :], m, 1, k)
            assert ok == 1 and rel_err(got[tk, th], c[0]) <= 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("t,m,k,n", [(T.Q4_K, 1024, 1024, 1), (T.Q6_K, 512, 512, 1), (T.Q4_K, 768, 512, 5), (T.Q8_0, 256, 256, 1), (T.F16, 128, 256, 2),
                                     (T.Q4_K, 512, 512, 20)], ids=lambda v: str(v))
def test_sibling_mul_mat_nodes_run_as_one_call(gpu, oracle, host_exe, tmp_path, t, m, k, n):
    """graph_compute (ggml-cuda.cu.patch:18945) sees attn_q / attn_k / attn_v and ffn_gate / ffn_up back to back over one src1:
    up to four such MUL_MAT nodes with at most 8 activation rows go out as ONE lfamd_mul_mat_multi_types call.  The host
    program puts a second node with its own copy of the weights behind the first and requires identical bytes from both;
    the result is held against the oracle as for a single node; LFAMD_BACKEND_NO_SIBLING_FUSION=1 runs them one by one."""
    W = synth.random_weights(t, m, k, 7)
    x = synth.random_activations(n, k, 8)
    wp, xp, op = tmp_path / "w.bin", tmp_path / "x.bin", tmp_path / "o.bin"
    W.tofile(wp)
    x.tofile(xp)
    outs = []
    for env in ({}, {"LFAMD_BACKEND_NO_SIBLING_FUSION": "1"}):
        r = subprocess.run([host_exe, _hip.HIP_SO, "mulmat", str(t), str(m), str(k), str(n), "1", str(wp), str(xp), str(op)],
                           capture_output=True, text=True, timeout=300,
                           env={**os.environ, "BACKEND_HOST_PAIR": "1", "LFAMD_BACKEND_STATS": "1", **env})
        assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr
        # (the host program computes the graph four times; more than 8 rows: no fusion, the batch bodies take one matrix each)
        assert f"{0 if env or n > 8 else 4} sibling calls" in r.stderr, r.stderr
        outs.append(np.fromfile(op, dtype=np.float32).reshape(n, m))
        check_mul_mat(oracle, outs[-1], t, W, x, m, k, n, 1)
    if n == 1:
        assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))  # one column: the fused launch is the same arithmetic


@pytest.mark.gpu
@pytest.mark.parametrize("tokens,weights_usage", [(1, True), (3, True), (12, True), (3, False)])
def test_sibling_mul_mat_id_nodes_run_as_one_call(gpu, oracle, host_exe, tmp_path, tokens, weights_usage):
    """ffn_gate_exps and ffn_up_exps of a layer (same src1, same ids) behind each other: one lfamd_mul_mat_id_multi call up to 4 tokens."""
    t, m, k, experts, thinkers, tasks = T.Q4_K, 64, 512, 6, 2, 1
    W = np.stack([synth.random_weights(t, m, k, 50 + e) for e in range(experts)])
    x = synth.random_activations(tokens * tasks, k, 9).reshape(tokens, tasks, k)
    rng = np.random.default_rng(4)
    ids = np.stack([rng.permutation(experts)[:thinkers] for _ in range(tokens)]).astype(np.int32)
    wp, xp, ip, op = (tmp_path / n for n in ("w.bin", "x.bin", "i.bin", "o.bin"))
    W.tofile(wp), x.tofile(xp), ids.tofile(ip)
    r = subprocess.run([host_exe, _hip.HIP_SO, "mulmatid", str(t), str(m), str(k), str(experts), str(thinkers), str(tasks), str(tokens),
                        str(wp), str(xp), str(ip), str(op)], capture_output=True, text=True, timeout=300,
                       env={**os.environ, "BACKEND_HOST_PAIR": "1", "LFAMD_BACKEND_STATS": "1",
                            **({} if weights_usage else {"BACKEND_HOST_NO_WEIGHTS_USAGE": "1"})})
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr
    # (expert stacks outside a weights buffer share the context's one scratch image: such nodes must run one by one)
    assert f"{4 if tokens <= 4 and weights_usage else 0} sibling calls" in r.stderr, r.stderr
    got = np.fromfile(op, dtype=np.float32).reshape(tokens, thinkers, m)
    q = oracle.quantize(T.Q8_K, x.reshape(-1, k))
    G = np.zeros_like(got)
    for tk in range(tokens):
        for th in range(thinkers):
            ok, c = oracle.sgemm(t, W[ids[tk, th]], T.Q8_K, q[tk * tasks + th % tasks][None, :], m, 1, k)
            assert ok == 1
            G[tk, th] = c[0]
    assert rel_err(got, G) <= (1e-3 if tokens > 4 else 2e-6)
