"""Instruction-level invariant of the pivot pipeline, checked on the device assembly of the product build
(scripts/check_pivot_waitcnt.py; __graft_entry__.build() produces ransac_slam_amd/_obj/rel_kernels.s and runs the same check)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _checker():
    spec = importlib.util.spec_from_file_location("check_pivot_waitcnt", os.path.join(ROOT, "scripts", "check_pivot_waitcnt.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


LISTING = """
_ZN5rslam23sweep_persistent_kernelILi12EEEvPdl: ; @_ZN5rslam23sweep_persistent_kernelILi12EEEvPdl
	s_load_dwordx2 s[0:1], s[4:5], 0x0
	;;#ASMSTART
	s_mov_b32 m0, s34
	s_nop 0
	global_load_lds_dwordx4 v[6:7], off sc1
	;;#ASMEND
	v_add_f64 v[0:1], v[2:3], v[4:5]
%s
	ds_read_b64 v[8:9], v10
	v_mfma_f64_16x16x4_f64 v[0:7], v[8:9], v[8:9], v[0:7]
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	s_barrier
	s_waitcnt vmcnt(0)
	ds_read_b64 v[8:9], v10
	s_endpgm
.Lfunc_end0:
"""


def test_checker_sees_a_compiler_wait_in_front_of_an_lds_read():
    chk = _checker()
    bad = chk.check_lines((LISTING % "\ts_waitcnt vmcnt(0) lgkmcnt(1)").split("\n"))
    assert bad["functions"] == 1 and bad["dma_blocks"] == 1 and len(bad["violations"]) == 1
    # a counted wait, a wait for LDS only, the hand-written wait in front of the barrier and a wait behind it are all fine
    for ok in ("\ts_waitcnt vmcnt(2)", "\ts_waitcnt lgkmcnt(0)", "\tv_mov_b32 v0, v1"):
        good = chk.check_lines((LISTING % ok).split("\n"))
        assert good["dma_blocks"] == 1 and good["violations"] == []


def test_product_assembly_has_no_global_wait_inside_the_pivot_steps():
    asm = os.path.join(ROOT, "ransac_slam_amd", "_obj", "rel_kernels.s")
    if not os.path.exists(asm):
        pytest.skip("no device assembly in this copy of the tree (it is made by __graft_entry__.build(), which asserts the same)")
    rep = _checker().check(asm)
    assert rep["functions"] >= 4 and rep["dma_blocks"] > 0
    assert rep["violations"] == []
