"""tools/check_valu_hazards.py on synthetic assembly: the software-managed VALU hazards of the gfx90a / gfx940 family at text level (the
compiler's hazard recogniser does not see into an inline-asm instruction).  The build runs the tool on every unit's
device assembly (nuts_rs_amd/build.py)."""
import os
import subprocess
import sys

TOOL = os.path.join(os.path.dirname(__file__), "..", "tools", "check_valu_hazards.py")


def scan(tmp_path, body):
    f = tmp_path / "k.s"
    f.write_text("_Z6kernelv:\n" + "".join("\t" + l + "\n" for l in body))
    r = subprocess.run([sys.executable, TOOL, str(f)], capture_output=True, text=True)
    return r.returncode, r.stdout


def test_dpp_needs_two_wait_states(tmp_path):
    dpp = "v_mov_b32_dpp v2, v0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
    rc, out = scan(tmp_path, ["v_add_f64 v[0:1], v[0:1], v[2:3]", dpp])
    assert rc == 1 and "'A': 1" in out
    assert scan(tmp_path, ["v_add_f64 v[0:1], v[0:1], v[2:3]", "s_nop 0", dpp])[0] == 1
    assert scan(tmp_path, ["v_add_f64 v[0:1], v[0:1], v[2:3]", "s_nop 1", dpp])[0] == 0
    assert scan(tmp_path, ["v_add_f64 v[0:1], v[0:1], v[2:3]", "v_mov_b32_e32 v9, v8", "v_mov_b32_e32 v10, v8", dpp])[0] == 0
    assert scan(tmp_path, ["v_add_f64 v[4:5], v[0:1], v[2:3]", dpp])[0] == 0          # another register


def test_readlane_and_permlane_swap(tmp_path):
    rc, out = scan(tmp_path, ["v_fma_f64 v[0:1], v[2:3], v[4:5], v[6:7]", "v_readlane_b32 s1, v1, 0"])
    assert rc == 1 and "'B': 1" in out
    assert scan(tmp_path, ["v_fma_f64 v[0:1], v[2:3], v[4:5], v[6:7]", "s_nop 0", "v_readlane_b32 s1, v1, 0"])[0] == 0
    rc, out = scan(tmp_path, ["v_mov_b32_e32 v2, v0", "v_permlane16_swap_b32_e32 v0, v2"])
    assert rc == 1 and "'C': 1" in out
    assert scan(tmp_path, ["v_mov_b32_e32 v2, v0", "v_mov_b32_e32 v3, v1", "s_nop 0", "v_permlane16_swap_b32_e32 v0, v2"])[0] == 0


def test_sgpr_forwarding_and_second_destinations(tmp_path):
    rc, out = scan(tmp_path, ["v_readlane_b32 s3, v254, 3", "v_bfrev_b32_e32 v26, 1", "v_cndmask_b32_e64 v23, v26, v21, s[2:3]"])
    assert rc == 1 and "'D': 1" in out
    assert scan(tmp_path, ["v_readlane_b32 s3, v254, 3", "v_bfrev_b32_e32 v26, 1", "v_mul_f64 v[18:19], v[16:17], v[18:19]",
                           "v_cndmask_b32_e64 v23, v26, v21, s[2:3]"])[0] == 0
    # the second operand of v_mad_u64_u32 is its carry-out DESTINATION: writing it right after a v_readlane of the same SGPR is no hazard
    assert scan(tmp_path, ["v_readlane_b32 s4, v252, 56", "v_mov_b32_e32 v9, v8", "v_mad_u64_u32 v[0:1], s[4:5], s4, v210, v[108:109]"])[0] == 1   # (s4 IS read)
    assert scan(tmp_path, ["v_readlane_b32 s5, v252, 56", "v_mov_b32_e32 v9, v8", "v_mad_u64_u32 v[0:1], s[4:5], s8, v210, v[108:109]"])[0] == 0


def test_transcendental_forwarding_and_labels(tmp_path):
    rc, out = scan(tmp_path, ["v_rcp_f64_e32 v[0:1], v[2:3]", "v_mul_f64 v[4:5], v[0:1], v[6:7]"])
    assert rc == 1 and "'F': 1" in out
    assert scan(tmp_path, ["v_rcp_f64_e32 v[0:1], v[2:3]", "s_nop 0", "v_mul_f64 v[4:5], v[0:1], v[6:7]"])[0] == 0
    # a local label does not reset the window (the fall-through path is a path); the next function's label does
    assert scan(tmp_path, ["v_rcp_f64_e32 v[0:1], v[2:3]", ".LBB0_1:", "v_mul_f64 v[4:5], v[0:1], v[6:7]"])[0] == 1
    assert scan(tmp_path, ["v_rcp_f64_e32 v[0:1], v[2:3]", "_Z5otherv:", "v_mul_f64 v[4:5], v[0:1], v[6:7]"])[0] == 0


def test_an_unconditional_branch_ends_the_window(tmp_path):
    """What follows an `s_branch` in the TEXT is another block, not the next instruction executed (round 6: the scan paired a v_readlane in front of
    a branch with the first load of the block printed behind it and failed a correct build)."""
    vm = "global_load_dwordx2 v[80:81], v47, s[0:1]"
    rc, out = scan(tmp_path, ["v_readlane_b32 s1, v252, 53", vm])
    assert rc == 1 and "'G': 1" in out                                      # the hazard itself is still seen in straight-line code
    assert scan(tmp_path, ["v_readlane_b32 s1, v252, 53", "s_branch .LBB0_9", vm])[0] == 0
    assert scan(tmp_path, ["v_readlane_b32 s1, v252, 53", "s_cbranch_vccnz .LBB0_9", vm])[0] == 1      # a conditional branch may fall through
