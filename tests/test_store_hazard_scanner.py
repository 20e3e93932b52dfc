"""tools/check_store_hazard.py on synthetic assembly: the scanner must flag a buffer store whose data VGPRs are written
inside the hazard window and accept the guarded forms — for the 128-bit SGPR-offset stores of the wavefront kernels and, with --mubuf64,
for the 64-bit stores of the lane kernel (round 3)."""
import os
import subprocess
import sys

TOOL = os.path.join(os.path.dirname(__file__), "..", "tools", "check_store_hazard.py")


def scan(tmp_path, body, *flags):
    f = tmp_path / "k.s"
    f.write_text("_Z6kernelv:\n" + "".join("\t" + l + "\n" for l in body))
    r = subprocess.run([sys.executable, TOOL, str(f), *flags], capture_output=True, text=True)
    return r.returncode, r.stdout


def test_128_bit_store_with_sgpr_offset(tmp_path):
    bad = ["buffer_store_dwordx4 v[4:7], v114, s[68:71], s8 offen", "v_fma_f64 v[4:5], v[10:11], v[12:13], v[14:15]"]
    rc, out = scan(tmp_path, bad)
    assert rc == 1 and "1 unguarded" in out
    good = ["buffer_store_dwordx4 v[4:7], v114, s[68:71], s8 offen", "s_nop 1", "v_fma_f64 v[4:5], v[10:11], v[12:13], v[14:15]"]
    assert scan(tmp_path, good)[0] == 0
    other_regs = ["buffer_store_dwordx4 v[4:7], v114, s[68:71], s8 offen", "v_fma_f64 v[8:9], v[10:11], v[12:13], v[14:15]"]
    assert scan(tmp_path, other_regs)[0] == 0
    imm_offset = ["buffer_store_dwordx4 v[4:7], v114, s[68:71], 0 offen", "v_fma_f64 v[4:5], v[10:11], v[12:13], v[14:15]"]
    assert scan(tmp_path, imm_offset)[0] == 0          # LLVM's own rule covers the immediate-soffset form


def test_64_bit_mubuf_store(tmp_path):
    bad = ["buffer_store_dwordx2 v[0:1], v3, s[0:3], 0 offen", "v_lshl_add_u64 v[0:1], v[6:7], 0, s[6:7]"]
    assert scan(tmp_path, bad)[0] == 0                 # not looked at without the flag
    rc, out = scan(tmp_path, bad, "--mubuf64")
    assert rc == 1 and "1 unguarded" in out
    # the guard keeps the data alive; the scheduler may still put the next offset's read directly behind the store
    good = ["buffer_store_dwordx2 v[226:227], v1, s[76:79], 0 offen", "v_accvgpr_read_b32 v1, a42", "s_nop 1", "v_mov_b64_e32 v[226:227], 0"]
    assert scan(tmp_path, good, "--mubuf64")[0] == 0
