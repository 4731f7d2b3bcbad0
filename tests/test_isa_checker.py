"""
The build-time ISA check of the batched CTM kernel (tools/check_smem_inflight.py, run by topicmodelsvb.jl_amd._lib.build):
it must flag a read of a scalar-load destination that is still in flight and a compiler write to the reserved SGPRs inside
a fixed-register streaming region, and pass clean code.  No GPU needed.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = os.path.join(ROOT, "tools", "check_smem_inflight.py")


def run(tmp_path, body, want="ctm_estep_batch"):
    p = tmp_path / "k.s"
    p.write_text("_Z22ctm_estep_batch_kernelILi52ELb0EEv12CtmBatchArgs: ; @kernel\n" + body + "\n\ts_endpgm\n")
    r = subprocess.run([sys.executable, CHECK, str(p), want], capture_output=True, text=True)
    return r.returncode, r.stdout


def test_clean_pipeline_passes(tmp_path):
    rc, out = run(tmp_path, """
	; CBFX_BEGIN
	s_mov_b64 s[34:35], s[4:5]
	s_load_dwordx16 s[36:51], s[34:35], 0
	s_waitcnt lgkmcnt(0)
	s_load_dwordx16 s[68:83], s[34:35], 0x80
	v_pk_fma_f32 v[2:3], s[36:37], v[10:11], v[2:3] op_sel_hi:[1,0,1]
	s_waitcnt lgkmcnt(0)
	v_pk_fma_f32 v[2:3], s[68:69], v[10:11], v[2:3] op_sel_hi:[1,0,1]
	; CBFX_END""")
    assert rc == 0 and "0 hazards (1 fixed-register streaming regions checked)" in out


def test_read_of_in_flight_destination_is_flagged(tmp_path):
    rc, out = run(tmp_path, """
	s_load_dwordx16 s[4:19], s[0:1], 0
	v_cvt_f64_f32_e32 v[18:19], s4
	s_waitcnt lgkmcnt(0)""")
    assert rc == 1 and "reads in-flight SGPR" in out
    rc, out = run(tmp_path, """
	s_load_dwordx16 s[52:67], s[20:21], 0x40
	v_writelane_b32 v255, s53, 2
	s_waitcnt vmcnt(0) lgkmcnt(0)""")
    assert rc == 1                                   # a spill of an in-flight group


def test_partial_wait_does_not_clear_in_flight_state(tmp_path):
    rc, _ = run(tmp_path, """
	s_load_dwordx4 s[8:11], s[0:1], 0
	s_waitcnt lgkmcnt(1)
	v_mov_b32_e32 v0, s9
	s_waitcnt lgkmcnt(0)""")
    assert rc == 1                                   # SMEM returns out of order: only lgkmcnt(0) is a completion point


def test_compiler_write_inside_streaming_region_is_flagged(tmp_path):
    rc, out = run(tmp_path, """
	; CBFX_BEGIN
	s_load_dwordx16 s[36:51], s[4:5], 0
	s_waitcnt lgkmcnt(0)
	s_or_saveexec_b64 s[40:41], -1
	v_pk_fma_f32 v[2:3], s[42:43], v[10:11], v[2:3] op_sel_hi:[1,0,1]
	; CBFX_END""")
    assert rc == 1 and "writes a reserved SGPR" in out
    rc, out = run(tmp_path, """
	; CBFX_BEGIN
	s_load_dwordx16 s[68:83], s[4:5], 0
	s_waitcnt lgkmcnt(0)
	v_readlane_b32 s70, v254, 3
	v_pk_fma_f32 v[2:3], s[68:69], v[10:11], v[2:3] op_sel_hi:[1,0,1]
	; CBFX_END""")
    assert rc == 1                                   # a spill reload into the block between its load and its last FMA


def test_block_is_free_after_its_last_fma_and_before_its_load(tmp_path):
    rc, out = run(tmp_path, """
	; CBFX_BEGIN
	s_mov_b64 s[36:37], s[4:5]
	s_load_dwordx16 s[36:51], s[4:5], 0
	s_waitcnt lgkmcnt(0)
	v_pk_fma_f32 v[2:3], s[50:51], v[10:11], v[2:3] op_sel_hi:[1,0,1]
	s_mov_b64 s[40:41], vcc
	s_load_dwordx16 s[36:51], s[4:5], 0x100
	s_waitcnt lgkmcnt(0)
	v_pk_fma_f32 v[2:3], s[36:37], v[10:11], v[2:3] op_sel_hi:[1,0,1]
	; CBFX_END""")
    assert rc == 0, out


def test_other_kernels_are_ignored(tmp_path):
    p = tmp_path / "k.s"
    p.write_text("_Z9somethingv: ; @x\n\ts_load_dwordx2 s[4:5], s[0:1], 0\n\tv_mov_b32_e32 v0, s4\n\ts_waitcnt lgkmcnt(0)\n\ts_endpgm\n")
    r = subprocess.run([sys.executable, CHECK, str(p), "ctm_estep_batch"], capture_output=True, text=True)
    assert r.returncode == 0


def test_built_library_passed_the_check():
    """build() keeps the ISA it checked; when it is present (a build happened in this tree) it must be clean."""
    isa = os.path.join(ROOT, "topicmodelsvb.jl_amd", "build", "tmvb_ctm-hip-amdgcn-amd-amdhsa-gfx950.s")
    if not os.path.exists(isa):
        import pytest
        pytest.skip("no saved ISA in this tree")
    r = subprocess.run([sys.executable, CHECK, isa, "ctm_estep_batch"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-1500:]
    assert "streaming regions checked" in r.stdout and " 0 fixed-register" not in r.stdout


# ---- round 6: tools/check_vmem_inflight.py -- the hand-scheduled token loop of ctm_estep_quad_kernel (inline-asm global loads, counted s_waitcnt vmcnt)
VCHECK = os.path.join(ROOT, "tools", "check_vmem_inflight.py")

_LOOP = """
	; CQVM_BEGIN
	global_load_dword v244, v10, s[8:9]
	s_waitcnt vmcnt(0)
	global_load_dword v245, v10, s[8:9]
	global_load_dwordx4 v[0:3], v20, s[2:3] offset:0
	global_load_dwordx4 v[4:7], v20, s[2:3] offset:0
	global_load_dwordx4 v[8:11], v20, s[2:3] offset:0
	global_load_dwordx4 v[12:15], v20, s[2:3] offset:0
.LBB1_1:
	s_waitcnt vmcnt(3)
	v_mov_b32_e32 v30, v245
%s
	global_load_dword v245, v31, s[8:9]
	v_pk_fma_f32 v[40:41], v[0:1], v[16:17], v[40:41]
	global_load_dwordx4 v[0:3], v20, s[2:3] offset:0
	s_waitcnt vmcnt(4)
	v_pk_fma_f32 v[40:41], v[4:5], v[16:17], v[40:41]
	global_load_dwordx4 v[4:7], v20, s[2:3] offset:0
	s_waitcnt vmcnt(4)
	v_pk_fma_f32 v[40:41], v[8:9], v[16:17], v[40:41]
	global_load_dwordx4 v[8:11], v20, s[2:3] offset:0
	s_waitcnt vmcnt(4)
	v_pk_fma_f32 v[40:41], v[12:13], v[16:17], v[40:41]
	global_load_dwordx4 v[12:15], v20, s[2:3] offset:0
	s_cbranch_scc0 .LBB1_1
	s_waitcnt vmcnt(0)
	; CQVM_END"""


def vrun(tmp_path, body):
    p = tmp_path / "q.s"
    p.write_text("\t.text\n_Z21ctm_estep_quad_kernelILi4ELb0EEv12CtmBatchArgs: ; @kernel\n" + body + "\n\ts_endpgm\n\t.end_amdhsa_kernel\n")
    r = subprocess.run([sys.executable, VCHECK, str(p)], capture_output=True, text=True)
    return r.returncode, r.stdout


def test_vmem_clean_hand_scheduled_loop_passes(tmp_path):
    rc, out = vrun(tmp_path, _LOOP % "\tv_add_u32_e32 v31, 4, v31")
    assert rc == 0 and "0 finding(s)" in out, out


def test_vmem_touch_of_an_in_flight_destination_is_flagged(tmp_path):
    """what the compiler did during round 6: the id word's register (still in flight from the previous round) used as an address temporary in front of the wait"""
    bad = _LOOP.replace("\ts_waitcnt vmcnt(3)\n\tv_mov_b32_e32 v30, v245", "\tv_mov_b32_e32 v30, v245\n\tv_add_u32_e32 v245, 4, v31\n\ts_waitcnt vmcnt(3)")
    rc, out = vrun(tmp_path, bad % "\tv_add_u32_e32 v31, 4, v31")
    assert rc == 1 and "is in flight" in out and "v245" in out.replace("[245]", "v245"), out


def test_vmem_kernel_without_markers_is_not_checked(tmp_path):
    rc, out = vrun(tmp_path, "\tglobal_load_dwordx4 v[0:3], v20, s[2:3]\n\tv_mov_b32_e32 v1, v0")
    assert rc == 0 and "nothing to check" in out
