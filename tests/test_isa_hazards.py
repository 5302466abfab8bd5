"""CPU: the ISA of the inline-asm GEMM kernels never touches a register a load is still in flight to (tools/isa_hazards.py).
hipcc cross-compiles gfx950 here (device code only, a few seconds per translation unit)."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_hazards  # noqa: E402

TUS = ["gemm_lw.hip", "gemm_mfma.hip", "gemm_wide_k4.hip", "gemm_wide_k5.hip", "gemm_wide_k6.hip", "gemm_wide_l4.hip",
       "gemm_wide_l5.hip", "gemm_wide_c16.hip", "gemm_wide_misc.hip"]  # every translation unit with inline-asm loads


@pytest.mark.skipif(not os.path.exists(isa_hazards.HIPCC), reason="needs hipcc")
def test_no_register_is_touched_before_its_load_is_waited_for():
    with ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(lambda f: isa_hazards.check_file(os.path.join(ROOT, "llamafile_amd", "csrc", f)), TUS))
    for f, res in zip(TUS, results):
        assert res, f  # kernels were found
        for kernel, bad in res.items():
            assert not bad, (f, kernel, bad[:3])


KS_TUS = [("gemm_ks.hip", nb) for nb in (1, 5, 8)] + [("gemm_kr.hip", nb) for nb in (1, 3, 4)] + [("gemm_i8.hip", nb) for nb in (1, 2, 4)] + \
         [("gemm_lf.hip", nb) for nb in (1, 3, 4)]


@pytest.mark.skipif(not os.path.exists(isa_hazards.HIPCC), reason="needs hipcc")
def test_k_split_and_row_split_bodies():
    """gemm_ks / gemm_kr / gemm_i8 / gemm_lf keep loads in flight across several periods of a rolled loop whose blocks hipcc lays out out of
    execution order; the linear walk needs straight-line code, which a fixed trip count (-DKS_CHECK_NB) gives.  The
    VALU-written-SGPR rule is position-local and runs on the shipped (rolled) build as well."""
    def one(arg):
        f, nb = arg
        return isa_hazards.check_file(os.path.join(ROOT, "llamafile_amd", "csrc", f), (f"-DKS_CHECK_NB={nb}", f"-DI8_CHECK_NB={nb}", f"-DLF_CHECK_NQ={nb}"))
    with ThreadPoolExecutor(max_workers=3) as ex:
        results = list(ex.map(one, KS_TUS))
    for (f, nb), res in zip(KS_TUS, results):
        assert res, f
        for kernel, bad in res.items():
            assert not bad, (f, nb, kernel, bad[:3])
    import subprocess
    import tempfile
    for f in ("gemm_ks.hip", "gemm_kr.hip", "gemm_i8.hip", "gemm_lf.hip"):
        with tempfile.NamedTemporaryFile(suffix=".s") as t:
            subprocess.run([isa_hazards.HIPCC, *isa_hazards.FLAGS, os.path.join(ROOT, "llamafile_amd", "csrc", f), "-o", t.name],
                           check=True, stderr=subprocess.DEVNULL)
            for kernel, bad in isa_hazards.check_sgpr_vmem(open(t.name).read()).items():
                assert not bad, (f, kernel, bad[:3])


def test_checker_flags_a_valu_written_sgpr_in_front_of_vmem():
    asm = """
_Z4demov: ; @demo
	v_readfirstlane_b32 s45, v3
	v_readfirstlane_b32 s44, v2
	global_load_dwordx4 v[10:13], v4, s[44:45] offset:0
	s_waitcnt vmcnt(0)
.Lfunc_end0:
"""
    assert isa_hazards.check_sgpr_vmem(asm)["_Z4demov"] == ["global_load_dwordx4 v[10:13], v4, s[44:45] offset:0"]
    ok = asm.replace("v_readfirstlane_b32 s44, v2\n", "v_readfirstlane_b32 s44, v2\n\ts_nop 4\n")
    assert isa_hazards.check_sgpr_vmem(ok)["_Z4demov"] == []
    other = asm.replace("s[44:45]", "s[46:47]")
    assert isa_hazards.check_sgpr_vmem(other)["_Z4demov"] == []


def test_checker_flags_a_copy_before_the_wait():
    asm = """
_Z4demov: ; @demo
	ds_read_b128 v[10:13], v2
	v_mov_b64_e32 v[20:21], v[10:11]
	s_waitcnt lgkmcnt(0)
	v_add_u32_e32 v3, v12, v13
.Lfunc_end0:
"""
    assert isa_hazards.check_asm(asm)["_Z4demov"] == ["v_mov_b64_e32 v[20:21], v[10:11]"]
    ok = asm.replace("v_mov_b64_e32 v[20:21], v[10:11]\n\ts_waitcnt lgkmcnt(0)", "s_waitcnt lgkmcnt(0)\n\tv_mov_b64_e32 v[20:21], v[10:11]")
    assert isa_hazards.check_asm(ok)["_Z4demov"] == []


GEMV_TUS = ["gemv_q4k.hip", "gemv_q5k.hip", "gemv_q6k.hip", "gemv_q40.hip", "gemv_q41.hip", "gemv_q50.hip", "gemv_q51.hip", "gemv_q2k.hip", "gemv_q3k.hip", "gemv_iq4xs.hip",
            "gemv_dual.hip"]
# (the inline-asm VALU sites — the `v_permlane*_swap` pair — live in gemv_impl.h, shared by every gemv_*.hip)
VALU_TUS = GEMV_TUS + ["gemv_q80.hip", "gemm_q80.hip", "moe.hip", "norm_quant.hip", "quantize.hip"]
SHIPPED_FLAGS = ("-mllvm", "-amdgpu-kernarg-preload-count=13")  # csrc/Makefile


@pytest.fixture(scope="module")
def shipped_asm():
    """Assembly of the shipped build of every unit the two tests below look at, compiled once (8 at a time)."""
    with ThreadPoolExecutor(max_workers=8) as ex:
        texts = list(ex.map(lambda f: isa_hazards.shipped_asm(os.path.join(ROOT, "llamafile_amd", "csrc", f), SHIPPED_FLAGS), VALU_TUS))
    return dict(zip(VALU_TUS, texts))


@pytest.mark.skipif(not os.path.exists(isa_hazards.HIPCC), reason="needs hipcc")
def test_decode_kernels_have_no_flat_access_and_no_stack(shipped_asm):
    """hipcc counts vmcnt exactly only without FLAT instructions and without a stack frame in the kernel; either one turned
    every counted wait of the decode GEMV into vmcnt(0) (no weight prefetch) while all parity tests stayed green."""
    for f in GEMV_TUS:
        res = isa_hazards.decode_hygiene(shipped_asm[f])
        decode = {k: v for k, v in res.items() if "gemv_kq" in k and ("Li1ELi" in k or "dual" in k)}  # NC = 1 bodies
        assert decode, f
        for kernel, probs in decode.items():
            assert not probs, (f, kernel[:80], probs)


@pytest.mark.skipif(not os.path.exists(isa_hazards.HIPCC), reason="needs hipcc")
def test_valu_wait_states_in_the_decode_and_exact_units(shipped_asm):
    """Every software-wait-state rule of tools/isa_hazards.py: check_valu_hazards over the shipped build of the decode GEMVs (the
    `v_permlane*_swap` pair of gemv_impl.h is inline asm), the exact Q8_0 batch kernel, the expert path, the quantisers."""
    for f in VALU_TUS:
        res = isa_hazards.check_valu_hazards(shipped_asm[f])
        assert res, f
        for kernel, bad in res.items():
            assert not bad, (f, kernel, bad[:3])


def test_checker_flags_valu_wait_state_violations():
    def run(body):
        return isa_hazards.check_valu_hazards("_Z4demov: ; @demo\n" + body + ".Lfunc_end0:\n")["_Z4demov"]
    # the permlane swap pair of gemv_impl.h without / with its s_nop 1
    assert len(run("\tv_add_f32_e32 v3, v1, v2\n\tv_permlane16_swap_b32_e32 v3, v4\n")) == 1
    assert len(run("\tv_add_f32_e32 v3, v1, v2\n\ts_nop 0\n\tv_permlane16_swap_b32_e32 v3, v4\n")) == 1
    assert run("\tv_add_f32_e32 v3, v1, v2\n\ts_nop 1\n\tv_permlane16_swap_b32_e32 v3, v4\n") == []
    assert run("\tv_add_f32_e32 v3, v1, v2\n\tv_mov_b32_e32 v9, v8\n\tv_mov_b32_e32 v10, v8\n\tv_permlane32_swap_b32_e32 v4, v3\n") == []
    assert run("\tv_add_f32_e32 v5, v1, v2\n\tv_permlane16_swap_b32_e32 v3, v4\n") == []  # another register
    # DPP source written by the previous VALU instruction; EXEC written in front of a DPP instruction
    assert len(run("\tv_max_f32_e32 v1, v1, v2\n\tv_mov_b32_dpp v3, v1 row_shr:1 row_mask:0xf bank_mask:0xf\n")) == 1
    assert run("\tv_max_f32_e32 v1, v1, v2\n\ts_nop 1\n\tv_mov_b32_dpp v3, v1 row_shr:1 row_mask:0xf bank_mask:0xf\n") == []
    assert len(run("\tv_cmpx_lt_f32_e32 v1, v2\n\ts_nop 3\n\tv_mov_b32_dpp v3, v7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")) == 1
    # readlane: source written by the previous VALU (1), lane select written by a VALU (4), SGPR result used as a VALU constant (2)
    assert len(run("\tv_add_u32_e32 v1, v1, v2\n\tv_readfirstlane_b32 s4, v1\n")) == 1
    assert run("\tv_add_u32_e32 v1, v1, v2\n\ts_nop 0\n\tv_readfirstlane_b32 s4, v1\n") == []
    assert len(run("\tv_readfirstlane_b32 s4, v1\n\ts_nop 2\n\tv_readlane_b32 s5, v2, s4\n")) == 1
    assert run("\tv_readfirstlane_b32 s4, v1\n\ts_nop 3\n\tv_readlane_b32 s5, v2, s4\n") == []
    assert len(run("\tv_readfirstlane_b32 s4, v1\n\tv_add_u32_e32 v2, s4, v3\n")) == 1
    assert run("\tv_readfirstlane_b32 s4, v1\n\ts_nop 1\n\tv_add_u32_e32 v2, s4, v3\n") == []
    assert run("\tv_mad_u64_u32 v[16:17], s[42:43], v17, v26, v[16:17]\n\tv_mad_u64_u32 v[18:19], s[42:43], v1, v2, v[16:17]\n") == []  # carry-out is a destination
    # transcendental result consumed by the next VALU; VCC in front of v_div_fmas; M0 in front of an LDS-DMA load
    assert len(run("\tv_rcp_f32_e32 v1, v2\n\tv_mul_f32_e32 v3, v1, v4\n")) == 1
    assert run("\tv_rcp_f32_e32 v1, v2\n\tv_mov_b32_e32 v9, v8\n\tv_mul_f32_e32 v3, v1, v4\n") == []
    assert len(run("\tv_cmp_lt_f32_e32 vcc, v1, v2\n\ts_nop 2\n\tv_div_fmas_f32 v3, v4, v5, v6\n")) == 1
    assert len(run("\ts_add_u32 m0, s3, 0x100\n\tglobal_load_lds_dwordx4 v1, s[4:5]\n")) == 1
    assert run("\ts_add_u32 m0, s3, 0x100\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v1, s[4:5]\n") == []
    # a packed 16-bit result read by the next VALU / MFMA instruction (gemm_lf.hip's dequantisation blocks: the multiply sits one
    # block behind the add it reads)
    assert len(run("\tv_pk_add_f16 v1, v2, v3\n\tv_pk_mul_f16 v4, v1, v5\n")) == 1
    assert run("\tv_pk_add_f16 v1, v2, v3\n\tv_perm_b32 v6, s2, v7, v8\n\tv_pk_mul_f16 v4, v1, v5\n") == []
    assert run("\tv_pk_add_f16 v1, v2, v3\n\ts_nop 0\n\tv_pk_mul_f16 v4, v1, v5\n") == []
    assert len(run("\tv_pk_mul_f16 v69, v69, v119\n\tv_mfma_f32_32x32x16_f16 v[50:65], v[82:85], v[66:69], v[50:65]\n")) == 1
    assert run("\tv_perm_b32 v1, s2, v7, v8\n\tv_pk_add_f16 v1, v2, v1\n") == []  # (the producer is not a packed instruction)
