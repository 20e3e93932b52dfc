"""tools/check_exec_spill.py: the scan (and repair) for the compiler defect behind DESIGN §22's wrong-results incidents.

The fixture is an excerpt of this repo's own compiler output — the one block of `nuts_draw_kernel<4,1,LrWrap<IidNormal>>` at 8e9c172 in which
hipcc 7.2.0 placed eleven VGPR spill stores ABOVE the join block's `s_or_b64 exec, exec, s[4:5]`; the edge that skips the `if` arrives with
EXEC == 0, so the stores store nothing (among them the main tree's log_size, the operand `merge_into` compares at src/nuts.rs:190-196)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "check_exec_spill.py")
FIXTURE = os.path.join(ROOT, "tests", "golden", "exec_spill_failing_block.s")


def _run(*args):
    return subprocess.run([sys.executable, TOOL] + list(args), capture_output=True, text=True)


def test_the_failing_block_is_flagged():
    r = _run(FIXTURE)
    assert r.returncode == 1
    assert "block .LBB6_681: 11 exec-dependent spill access(es)" in r.stdout and "[fixable]" in r.stdout
    assert "v[170:171], off offset:168" in _run(FIXTURE, "-v").stdout          # the main tree's log_size


def test_the_repair_moves_one_instruction_and_the_result_is_clean(tmp_path):
    out = tmp_path / "fixed.s"
    assert _run(FIXTURE, "--fix", str(out)).returncode == 0
    assert _run(str(out)).returncode == 0
    a = open(FIXTURE).read().split("\n")
    b = out.read_text().split("\n")
    assert len(a) == len(b)
    moved = [l for l in b if "moved up" in l]
    assert len(moved) == 1 and moved[0].strip().startswith("s_or_b64 exec, exec, s[4:5]")
    # everything else is the same text in the same order
    assert [l for l in a if "s_or_b64 exec, exec, s[4:5]" not in l] == [l for l in b if "s_or_b64 exec, exec, s[4:5]" not in l]
    # and the restore is now the block's first instruction
    k = b.index(next(l for l in b if l.startswith(".LBB6_681:")))
    first = next(l for l in b[k + 1:] if l.strip() and not l.strip().startswith(";"))
    assert first.strip().startswith("s_or_b64 exec, exec, s[4:5]")


def test_a_restore_that_cannot_be_moved_stays_an_error(tmp_path):
    # a VALU instruction between the label and the restore: not provably exec-independent -> the build must fail, not guess
    src = open(FIXTURE).read().replace("\tv_writelane_b32 v255, s65, 6\n", "\tv_writelane_b32 v255, s65, 6\n\tv_mov_b32_e32 v1, v2\n", 1)
    p = tmp_path / "unfixable.s"
    p.write_text(src)
    r = _run(str(p), "--fix", str(tmp_path / "o.s"))
    assert r.returncode == 1 and "[fixable]" not in r.stdout
    # ... and so does one whose SGPR pair is rewritten on the way
    src = open(FIXTURE).read().replace("\tv_writelane_b32 v255, s65, 6\n", "\tv_writelane_b32 v255, s65, 6\n\ts_mov_b64 s[4:5], s[8:9]\n", 1)
    p.write_text(src)
    assert _run(str(p), "--fix", str(tmp_path / "o.s")).returncode == 1


def test_spills_after_the_restore_or_in_blocks_not_entered_with_exec_zero_are_fine(tmp_path):
    ok = ("f:\n\ts_and_saveexec_b64 s[4:5], vcc\n\ts_cbranch_execz .LBB0_2\n\tv_mov_b32_e32 v0, 1\n.LBB0_2:\n\ts_or_b64 exec, exec, s[4:5]\n"
          "\tscratch_store_dword off, v0, off offset:4 ; 4-byte Folded Spill\n\ts_endpgm\n.Lfunc_end0:\n")
    p = tmp_path / "ok.s"
    p.write_text(ok)
    assert _run(str(p)).returncode == 0
    # the same stores inside the `if` body (a block that is not the target of the execz edge): the body's own exec is the right one
    body = ("f:\n\ts_and_saveexec_b64 s[4:5], vcc\n\ts_cbranch_execz .LBB0_2\n\tscratch_store_dword off, v0, off offset:4 ; 4-byte Folded Spill\n"
            "\ts_or_b64 exec, exec, s[4:5]\n.LBB0_2:\n\ts_endpgm\n.Lfunc_end0:\n")
    p.write_text(body)
    assert _run(str(p)).returncode == 0


@pytest.mark.skipif(not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_the_build_repairs_a_finding_and_replays_the_compiler_pipeline(tmp_path):
    """nuts_rs_amd.build._repair_exec_spills end to end, without a GPU: a small kernel's device assembly gets the defect's shape injected (SGPR-to-lane
    spill, VGPR spill, THEN the exec restore in a block entered by s_cbranch_execz); the build's hook must flag it, move the restore, re-run hipcc's own
    assembler / lld / bundler / host steps, and the OBJECT that comes out must carry the repaired order."""
    from nuts_rs_amd import build as B
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    src = tmp_path / "t.hip"
    src.write_text("#include <hip/hip_runtime.h>\n__global__ void k(double* p, int n) { int i = threadIdx.x; if (i < n) p[i] = p[i] * 2.0 + 1.0; }\n")
    obj = tmp_path / "t.o"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-save-temps=obj", "-c", str(src), "-o", str(obj)]
    subprocess.check_call(cmd, cwd=tmp_path)
    asm = next(p for p in tmp_path.iterdir() if p.name.endswith(".s") and "amdgcn" in p.name)
    text = asm.read_text()
    m = re.search(r"s_and_saveexec_b64 (s\[\d+:\d+\]), vcc\n\s*s_cbranch_execz (\.LBB0_\d+)\n", text)
    assert m, "the test kernel no longer compiles to a saveexec / execz skeleton"
    pair, label = m.group(1), m.group(2)
    inject = f"{label}:\n\tv_writelane_b32 v1, s0, 0\n\tscratch_store_dword off, v0, off offset:4\n\ts_or_b64 exec, exec, {pair}\n"
    assert text.count(f"{label}:\n") == 1
    asm.write_text(text.replace(f"{label}:\n", inject))
    assert _run(str(asm)).returncode == 1                                       # the injected shape is a finding
    B._repair_exec_spills(str(asm), cmd, "test kernel")
    assert _run(str(asm)).returncode == 0
    # the object hipcc's replayed pipeline produced: its device code has the restore in front of the spill store
    llvm = "/opt/rocm/lib/llvm/bin"
    subprocess.check_call([f"{llvm}/llvm-objcopy", f"--dump-section=.hip_fatbin={tmp_path / 'fb'}", str(obj)])
    subprocess.check_call([f"{llvm}/clang-offload-bundler", "-unbundle", "-type=o", "-targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"-input={tmp_path / 'fb'}",
                           f"-output={tmp_path / 'co'}"])
    dis = subprocess.check_output([f"{llvm}/llvm-objdump", "-d", "--no-show-raw-insn", str(tmp_path / "co")], text=True)
    i_or, i_st = dis.find("s_or_b64 exec, exec"), dis.find("scratch_store_dword")
    assert 0 < i_or < i_st, "the rebuilt object does not carry the repaired order"
