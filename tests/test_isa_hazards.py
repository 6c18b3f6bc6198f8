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


def test_the_scratch_checker_tells_a_spill_inside_a_loop_from_one_around_it():
    text = """
0000000000002000 <kern_a>:
	scratch_store_dword off, v1, off                           // 000000002000: DC000000
	v_add_f32_e32 v2, v2, v3                                   // 000000002008: 02040702
	s_cbranch_scc1 4093 <kern_a+0x8>                           // 00000000200C: BF85FFFE
	scratch_load_dword v1, off, off                            // 000000002010: DC000000
	s_cbranch_vccnz 4090 <kern_a+0x0>                          // 000000002018: BF87FFF9
	s_endpgm                                                   // 00000000201C: BF810000
0000000000003000 <kern_b>:
	v_add_f32_e32 v2, v2, v3                                   // 000000003000: 02040702
	scratch_load_dword v1, off, off                            // 000000003004: DC000000
	s_cbranch_scc1 4093 <kern_b+0x0>                           // 00000000300C: BF85FFFC
	s_endpgm                                                   // 000000003010: BF810000
"""
    assert isa_hazards.scratch_in_innermost_loops_of(text) == {"kern_a": (2, 0), "kern_b": (1, 1)}


def test_what_scratch_the_built_kernels_use_lies_around_their_loops():
    """every kernel compiles to zero scratch but the even-share form of the symmetric kernel at sixteen bodies per
    lane in the general form (values parked around the 64-step loops) and bh_small_build_kernel (parked between its phases) —
    and none of their scratch instructions sits inside an innermost loop."""
    seen = {}
    for obj in sorted(glob.glob(os.path.join(CSRC, "*.o"))):
        if ".variant-" in obj:
            continue
        try:
            found = isa_hazards.scratch_in_innermost_loops(obj)
        except RuntimeError:                                     # a host-only object (sym_plan.o, ic.o, actor.o): no device code inside
            continue
        for func, (n, inside) in found.items():
            seen[func] = (n, inside)
    assert seen, "build the library first (graft build())"
    for func, (n, inside) in seen.items():
        assert inside == 0, (func, n, inside)
        assert "bh_small_build_kernel" in func or ("forces_sym_pk_kernelILi8E" in func and func.split("EEvP")[0].endswith("Lb0ELb0ELb1E")), func
    assert any("bh_small_build_kernel" in f for f in seen) and sum("forces_sym_pk_kernel" in f for f in seen) <= 4
