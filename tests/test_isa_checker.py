"""tools/check_isa_hazards.py — the static check the Makefile runs on the traversal kernels' final ISA: nothing may touch
a register that a hand-issued (inline-asm) global_load is still writing.  Here: the checker itself on small synthetic
kernels (it must see a hazard where there is one, across branches and loop back edges, and none where there is not), and
on the ISA of the shipped build when the build directory is present."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "check_isa_hazards.py")


def run(tmp_path, body):
    f = tmp_path / "k.s"
    f.write_text("_Z6kernelv:\n" + body + "\ts_endpgm\n.Lfunc_end0:\n")
    r = subprocess.run([sys.executable, TOOL, str(f)], capture_output=True, text=True)
    return r.returncode, r.stdout


LOAD = "\t;;#ASMSTART\n\tglobal_load_dwordx4 v[10:13], v[2:3], off\n\t;;#ASMEND\n"
WAIT = "\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n"


def test_clean_sequence_passes(tmp_path):
    rc, out = run(tmp_path, LOAD + "\tv_add_f32_e32 v4, v5, v6\n" + WAIT + "\tv_mov_b32_e32 v7, v10\n")
    assert rc == 0 and "0 hazards" in out, out


def test_copy_of_an_in_flight_register_is_a_hazard(tmp_path):
    rc, out = run(tmp_path, LOAD + "\tv_mov_b32_e32 v7, v11\n" + WAIT)       # the phi-resolution copy of round 3's first build
    assert rc == 1 and "HAZARD" in out and "v[11]" in out.replace("v11", "v[11]"), out


def test_overwriting_an_in_flight_register_is_a_hazard(tmp_path):
    rc, out = run(tmp_path, LOAD + "\tv_mov_b32_e32 v12, 0\n" + WAIT)
    assert rc == 1, out


def test_address_in_a_register_still_in_flight_is_a_hazard(tmp_path):
    second = "\t;;#ASMSTART\n\tglobal_load_dwordx4 v[20:23], v[10:11], off\n\t;;#ASMEND\n"
    rc, out = run(tmp_path, LOAD + second + WAIT)
    assert rc == 1 and "address" in out, out


def test_hazard_behind_a_branch_and_around_a_loop_is_found(tmp_path):
    body = (LOAD + "\ts_cbranch_scc1 .LBB0_2\n\tv_add_f32_e32 v4, v5, v6\n.LBB0_2:\n\tv_mov_b32_e32 v7, v13\n" + WAIT)
    assert run(tmp_path, body)[0] == 1
    loop = (".LBB0_1:\n\tv_mov_b32_e32 v7, v10\n" + LOAD + "\ts_cbranch_scc1 .LBB0_1\n" + WAIT)   # the use is reached again with the load in flight
    assert run(tmp_path, loop)[0] == 1
    safe_loop = (".LBB0_1:\n" + LOAD + WAIT + "\tv_mov_b32_e32 v7, v10\n\ts_cbranch_scc1 .LBB0_1\n")
    assert run(tmp_path, safe_loop)[0] == 0


def test_a_compiler_waitcnt_lands_the_loads_too(tmp_path):
    rc, _ = run(tmp_path, LOAD + "\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\tv_mov_b32_e32 v7, v10\n")
    assert rc == 0


def test_counted_waits_land_everything_but_the_youngest(tmp_path):
    """round 4 (trav_coop): loads return in order — `s_waitcnt vmcnt(1)` behind two asm loads of one block lands the older"""
    second = "\t;;#ASMSTART\n\tglobal_load_dwordx4 v[20:23], v[2:3], off\n\t;;#ASMEND\n"
    counted = "\t;;#ASMSTART\n\ts_waitcnt vmcnt(1)\n\t;;#ASMEND\n"
    assert run(tmp_path, LOAD + second + counted + "\tv_mov_b32_e32 v7, v10\n" + WAIT)[0] == 0      # the older load has landed
    assert run(tmp_path, LOAD + second + counted + "\tv_mov_b32_e32 v7, v20\n" + WAIT)[0] == 1      # the younger one has not
    assert run(tmp_path, LOAD + ".LBB0_1:\n" + second + "\t;;#ASMSTART\n\ts_waitcnt vmcnt(2)\n\t;;#ASMEND\n\tv_mov_b32_e32 v7, v10\n" + WAIT)[0] == 1   # fewer than N in the block: nothing concluded


def test_a_branch_around_a_wait_inside_an_asm_statement_keeps_the_registers_in_flight(tmp_path):
    """numeric local labels of inline asm (`s_cbranch_scc1 6f` ... `6:`) are edges of the flow graph: the path that skips the
    wait reaches the use with the load still in flight; a branch that only skips LOADS is harmless"""
    skip_wait = LOAD + "\t;;#ASMSTART\n\ts_cmp_eq_u64 s[4:5], 0\n\ts_cbranch_scc1 6f\n\ts_waitcnt vmcnt(0)\n6:\n\t;;#ASMEND\n\tv_mov_b32_e32 v7, v10\n" + WAIT
    assert run(tmp_path, skip_wait)[0] == 1
    skip_load = ("\t;;#ASMSTART\n\ts_mov_b64 exec, s[4:5]\n\ts_cbranch_execz 2f\n\tglobal_load_dwordx4 v[10:13], v[2:3], off\n2:\n\t;;#ASMEND\n"
                 + "\tv_add_f32_e32 v4, v5, v6\n" + WAIT + "\tv_mov_b32_e32 v7, v10\n")
    assert run(tmp_path, skip_load)[0] == 0
    backwards = "3:\n" + LOAD + "\t;;#ASMSTART\n\ts_cbranch_scc1 3b\n\t;;#ASMEND\n" + WAIT       # the load's address register is read again with the load in flight: fine; its destination is not touched
    assert run(tmp_path, backwards)[0] == 0


def test_shipped_build_is_clean_when_present():
    files = sorted(glob.glob(os.path.join(ROOT, "audio-pathtracer_amd", "csrc", "build", "fs_*-hip-amdgcn-amd-amdhsa-gfx950.s")))
    files = [f for f in files if os.path.basename(f).split("-hip-")[0] in ("fs_walk", "fs_connect", "fs_frame", "fs_frame_wide", "fs_aux_kernels")]
    if not files:
        import pytest
        pytest.skip("no build directory (the Makefile keeps the ISA under csrc/build/)")
    r = subprocess.run([sys.executable, TOOL, *files], capture_output=True, text=True)
    assert r.returncode == 0 and " 0 hazards" in r.stdout, r.stdout[-2000:]
