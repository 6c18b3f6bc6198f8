"""The hand-written instructions of csrc/ (v_add_f32_dpp in kernels_block.hip; v_pk_mul_f32 op_sel and v_pk_fma_f32 clamp in
pk_common.h / kernels_sym.hip) sit outside LLVM's hazard recogniser.  tools/isa_hazards.py walks the disassembly of the BUILT
objects — the code that ships — for the gfx950 hazards the hardware does not interlock: a DPP read of a VGPR written by a VALU
instruction in the two slots before it, a DPP instruction within five slots of a VALU write of EXEC, a VALU read of a
transcendental result in the next slot."""
import glob
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_hazards  # noqa: E402

CSRC = os.path.join(ROOT, "parallelnbody_amd", "csrc")


def test_the_checker_sees_each_kind_of_hazard():
    text = """
0000000000001000 <kern>:
	v_add_f32_e32 v10, v92, v18                                // 000000001000: 0224255C
	v_add_f32_dpp v10, v10, v10 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf// 000000001004: 021414FA FF004E0A
	v_add_f32_e32 v11, v92, v18
	v_mov_b32_e32 v1, v2
	v_add_f32_dpp v11, v11, v11 row_mirror row_mask:0xf bank_mask:0xf
	v_add_f32_e32 v12, v92, v18
	s_nop 1
	v_add_f32_dpp v12, v12, v12 row_mirror row_mask:0xf bank_mask:0xf
	v_pk_fma_f32 v[20:21], v[2:3], v[4:5], v[6:7]
	v_mov_b32_dpp v3, v21 row_shr:1 row_mask:0xf bank_mask:0xf
	v_rsq_f32_e32 v5, v6
	v_pk_mul_f32 v[8:9], v[4:5], v[4:5]
	v_rsq_f32_e32 v5, v6
	s_nop 0
	v_mul_f32_e32 v7, v5, v5
	v_cmpx_gt_f32_e32 v1, v2
	v_mov_b32_e32 v1, v2
	v_mov_b32_e32 v1, v2
	v_mov_b32_dpp v3, v4 row_shr:1 row_mask:0xf bank_mask:0xf
"""
    found = [(kind, n) for _f, n, kind, _t in isa_hazards.check(text)]
    assert found == [("dpp", 4), ("dpp", 7), ("dpp", 12), ("trans", 14), ("exec", 21)]


def test_no_uninterlocked_hazard_in_any_built_kernel():
    objs = sorted(glob.glob(os.path.join(CSRC, "kernels*.o")))
    objs = [o for o in objs if ".variant-" not in o]
    assert len(objs) >= 5, "build the library first (graft build())"
    total_dpp = 0
    for obj in objs:
        found, n_dpp, _ = isa_hazards.check_object(obj)
        total_dpp += n_dpp
        assert not found, (obj, found[:5])
    assert total_dpp > 10000            # the written-out wave sums of forces_block_pk_kernel alone are ~10 000 v_add_f32_dpp


def test_the_written_out_dpp_adds_keep_their_distance():
    # what wave_sum_to_lane63 promises beyond the two wait states: consecutive steps on one register are NV >= 8 instructions apart
    found, n_dpp, _ = isa_hazards.check_object(os.path.join(CSRC, "kernels_block.o"))
    assert not found and n_dpp >= 8000
