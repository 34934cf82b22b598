"""Static fence for the cross-lane (DPP) read hazard: every `*_dpp` instruction of every gfx950 code object in the
SHIPPED library is checked -- its src0 (the register read from another lane) must not have been written by a VALU instruction fewer than
two wait states earlier.  The compiler keeps that distance for its own DPP moves; the stencils' inline-asm `v_fmac_f32_dpp` sequences
(csrc/wv_rb.hip, csrc/wv_h16.hip) keep it by opening every sequence with the wait inside the same asm statement.  CPU only: needs the
built library and llvm-objdump, no GPU."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import dpp_hazard  # noqa: E402


def test_the_checker_sees_a_hazard_when_there_is_one():
    asm = """
0000000000001000 <k>:
	v_mov_b32_e32 v5, v9                                        // 000000001000: 7E0A0309
	v_fmac_f32_dpp v1, v5, v2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1 // 000000001004: 00000000
	v_add_f32_e32 v7, v1, v1                                    // 000000001008: 00000000
	s_nop 0                                                     // 00000000100c: BF800000
	v_mov_b32_dpp v8, v7 wave_shl:1 row_mask:0xf bank_mask:0xf  // 000000001010: 00000000
	v_mul_f32_e32 v[10:11], v1, v1                              // 000000001018: 00000000
	s_nop 1                                                     // 00000000101c: BF800001
	v_fmac_f32_dpp v1, v11, v2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1 // 000000001020: 00000000
	v_mov_b32_e32 v12, v9                                       // 000000001028: 7E0A0309
	v_mov_b32_e32 v13, v9                                       // 00000000102c: 7E0A0309
	v_fmac_f32_dpp v1, v12, v2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1 // 000000001030: 00000000
"""
    n, hz = dpp_hazard.scan_asm(asm)
    assert n == 4
    # v5 written right before its DPP read (0 wait states); v7 with one s_nop 0 (1 wait state); v11 behind s_nop 1 is clean;
    # v12 with one instruction in between (1 wait state)
    assert [h["wait_states"] for h in hz] == [0, 1, 1], hz
    assert "v5" in hz[0]["dpp"] and "v7" in hz[1]["dpp"] and "v12" in hz[2]["dpp"]


def test_no_dpp_read_follows_its_writer_within_two_wait_states():
    from waveverify_amd.build import LIB
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    r = dpp_hazard.scan_library(LIB)
    assert r["code_objects"] >= 6 and r["dpp_instructions"] > 5000, r      # the stencil kernels alone hold thousands
    assert not r["hazards"], "\n".join(f"{h['function'][:80]}: {h['writer']}  ->  {h['dpp']}" for h in r["hazards"][:20])
