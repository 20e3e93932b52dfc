"""Build libnuts_amd.so (the HIP engine) in-tree for gfx950.  `python -m nuts_rs_amd.build [--force] [-v]`.

The kernels of each density live in their own translation unit (csrc/kern_*.hip) so the build runs in parallel;
every unit is compiled with hipcc --offload-arch=gfx950 (cross-compiles without a GPU) and the objects are linked
into one shared library that exports the C ABI of include/nuts_amd.h.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libnuts_amd.so")
# (the slowest translation units first: the build is as long as its longest chain of jobs on the available cores)
UNITS = ["kern_tile_mvn_prec.hip", "kern_tile_mvn_diag.hip", "kern_lockstep.hip", "kern_lr_mvn_prec.hip@large", "kern_lr_mvn_prec.hip@small",
         "kern_kin_mvn_prec.hip@large", "kern_kin_mvn_prec.hip@small", "kern_mvn_prec.hip@large", "kern_mvn_prec.hip@small", "kern_lr_iid_normal.hip@large",
         "kern_lr_iid_normal.hip@small", "kern_lr_diag_normal.hip@large", "kern_lr_diag_normal.hip@small", "kern_lr_funnel.hip@large", "kern_lr_funnel.hip@small",
         "kern_lr_host_cb.hip@large", "kern_lr_host_cb.hip@small", "kern_kin_iid_normal.hip@large", "kern_kin_iid_normal.hip@small", "kern_kin_diag_normal.hip@large",
         "kern_kin_diag_normal.hip@small", "kern_kin_funnel.hip@large", "kern_kin_funnel.hip@small", "kern_kin_host_cb.hip@large", "kern_kin_host_cb.hip@small", "kern_lane.hip",
         "kern_lane_kin.hip", "kern_iid_normal.hip@large", "kern_iid_normal.hip@small", "kern_diag_normal.hip@large", "kern_diag_normal.hip@small", "kern_funnel.hip@large",
         "kern_funnel.hip@small", "kern_host_cb.hip@large", "kern_host_cb.hip@small", "nuts_engine.hip", "kern_cluster.hip", "kern_cluster_kin.hip", "kern_eight_schools.hip@inl",
         "kern_lr_eight_schools.hip@inl", "kern_kin_eight_schools.hip@inl", "math_seam.hip", "probe_bw.hip", "pooled_reduce.hip", "lowrank_device.hip", "lowrank_host.cpp"]
# (lowrank_host.cpp holds both ISA builds of the host estimator in ONE translation unit: per-function target attributes, see there)
# Variants of a unit ("file@variant": its own object, the flags below).  A density's one-chain kernels are TWO units from one source
# (nuts_launch.hpp NM_TU_PART): "small" = the tilings of <= 4 doubles per lane + the small-chain kernels with EVERY special function inlined — no
# out-of-line device call (fourth incident; the scan below rejects an s_swappc_b64 there) —, "large" = the 8- and 16-doubles tilings with
# the calls (inlined they cost K2 11 %); "inl" = a unit that only has small tilings (8 schools).
VARIANT_FLAGS = {"": [], "small": ["-DNM_TU_PART=1", "-DNM_DETMATH_INLINE=1"], "large": ["-DNM_TU_PART=2"], "inl": ["-DNM_DETMATH_INLINE=1"]}
NO_CALL_VARIANTS = ("small", "inl")
HEADERS = ["nuts_kernels.hpp", "nuts_launch.hpp", "dev_math.hpp", "detmath_tables.hpp", "zig_tables.hpp", "nuts_group.hpp", "nuts_group_impl.hpp", "nuts_tile.hpp", os.path.join("..", "..", "include", "nuts_amd.h")]
# -ffp-contract=off: FMAs only where the reference writes mul_add
# -Wno-pass-failed: "loop not unrolled" remarks of the matrix-core kernel's partially unrolled product loops (a diagnostic only)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-pass-failed"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _includes(path, seen=None):
    """The quoted includes a source pulls in (recursively, resolved against csrc/ and include/): a unit is recompiled when
    one of ITS headers changed, not when any header did (nuts_tile.hpp concerns two units of 26)."""
    import re
    seen = set() if seen is None else seen
    try:
        text = open(path).read()
    except OSError:
        return seen
    for name in re.findall(r'^\s*#\s*include\s+"([^"]+)"', text, flags=re.M):
        for base in (os.path.dirname(path), CSRC, os.path.join(HERE, "..", "include")):
            q = os.path.normpath(os.path.join(base, name))
            if os.path.exists(q):
                if q not in seen:
                    seen.add(q)
                    _includes(q, seen)
                break
    return seen


def _src(u):
    return os.path.join(CSRC, u.split("@")[0])


def _obj(u):
    name, _, variant = u.partition("@")
    return os.path.join(OBJ, name.replace(".hip", "").replace(".cpp", "") + ("_" + variant if variant else "") + ".o")


def _unit_stale(u):
    src = _src(u)
    return _newer(_obj(u), [src] + sorted(_includes(src)))


def needs_build():
    """True when the library is missing, older than one of its objects, or an object is older than its source / its headers."""
    return any(_unit_stale(u) for u in UNITS) or _newer(LIB, [_obj(u) for u in UNITS])


class StoreHazardError(RuntimeError):
    """A translation unit's device assembly has a buffer store whose data registers are rewritten inside the hazard window."""


def _scan_store_hazards(asm_path, what, allow_calls=True):
    """tools/check_store_hazard.py's rule, both modes, on one unit's device assembly: the build fails on an unguarded
    instance instead of leaving it to a parity suite that does not travel with a user's module build.  The rule lives in ONE place: the
    tool is executed, not re-implemented (it ships in the repo next to this package; a wheel without tools/ skips the scan and says so)."""
    tool = os.path.join(HERE, "..", "tools", "check_store_hazard.py")
    if not os.path.exists(tool):
        print(f"nuts_rs_amd.build: tools/check_store_hazard.py not found, {what} NOT scanned for store-data hazards", file=sys.stderr)
        return
    for mode in ([], ["--mubuf64"]):
        r = subprocess.run([sys.executable, tool, asm_path] + mode, capture_output=True, text=True)
        if r.returncode != 0:
            raise StoreHazardError(f"{what}: unguarded store-data hazard(s) in the generated assembly (python tools/check_store_hazard.py "
                                   f"{' '.join(mode)}; guard the store as buf_store2 / LCtx::bst do, DESIGN §15):\n" + r.stdout[-3000:])
    # DESIGN §22's rule as a scan: no out-of-line call inside a kernel that carries several chains per wavefront
    tool2 = os.path.join(HERE, "..", "tools", "check_divergent_calls.py")
    if os.path.exists(tool2):
        r = subprocess.run([sys.executable, tool2, asm_path], capture_output=True, text=True)
        if r.returncode != 0:
            raise StoreHazardError(f"{what}: an out-of-line call inside a several-chains-per-wavefront kernel (python tools/check_divergent_calls.py; "
                                   "force-inline the callee, DESIGN §22):\n" + r.stdout[-3000:])


    # the software-managed VALU hazards (DPP / v_readlane / v_permlane*_swap / SGPR forwarding / transcendental results) at TEXT level: the
    # compiler's hazard recogniser cannot see into an inline-asm instruction (incidents three and four)
    tool3 = os.path.join(HERE, "..", "tools", "check_valu_hazards.py")
    if os.path.exists(tool3):
        r = subprocess.run([sys.executable, tool3, asm_path], capture_output=True, text=True)
        if r.returncode != 0:
            raise StoreHazardError(f"{what}: a software-managed VALU hazard without its wait states (python tools/check_valu_hazards.py):\n" + r.stdout[-3000:])


    # the policy of §22's fourth incident as a scan: no out-of-line device call in the units of the small tilings (and in user modules of them)
    if not allow_calls:
        with open(asm_path) as f:
            import re
            n_calls, callees, last_sym = 0, set(), None
            for line in f:
                m = re.search(r"s_add_u32\s+s\d+,\s*s\d+,\s*([\w.$]+)@rel32@lo", line)
                if m:
                    last_sym = m.group(1)
                if line.lstrip().startswith("s_swappc_b64"):
                    n_calls += 1
                    callees.add(last_sym or "?")
        if n_calls:
            raise StoreHazardError(f"{what}: {n_calls} out-of-line device call(s) (s_swappc_b64) to {sorted(callees)} in a unit that is built "
                                   "without them (-DNM_DETMATH_INLINE=1; DESIGN §22): mark the callee(s) __forceinline__ (a __noinline__ helper in a "
                                   "user density of a small tiling is not supported)")


class ExecSpillError(RuntimeError):
    """The compiler placed a VGPR spill / reload before the exec restore of a join block that is entered with EXEC == 0 (the root
    cause of the code-generation incidents of rounds 4 - 5) and the instance could not be repaired in the assembly."""


def _repair_exec_spills(asm_path, cmd, what):
    """tools/check_exec_spill.py on one unit's device assembly.  A finding is a miscompile of this toolchain (ROCm 7.2 hipcc: VGPR spill code
    inserted between a join block's hoisted SGPR-to-lane spills and its `s_or_b64 exec, exec, s[..]`; on the edge that skips the `if` with
    EXEC == 0 the spill stores nothing and the reload returns a stale slot).  The repair is done where the defect is: the exec restore is moved
    up to the block's first instruction in the ASSEMBLY (everything it jumps over is exec-independent or one of the spill accesses it is
    meant to cover), and the rest of the hipcc pipeline — assembler, lld, bundler, host compile — is replayed on the repaired file.  An
    instance the tool cannot prove movable fails the build."""
    import shlex
    tool = os.path.join(HERE, "..", "tools", "check_exec_spill.py")
    if not os.path.exists(tool):
        print(f"nuts_rs_amd.build: tools/check_exec_spill.py not found, {what} NOT scanned for spills before an exec restore", file=sys.stderr)
        return
    r = subprocess.run([sys.executable, tool, asm_path], capture_output=True, text=True)
    if r.returncode == 0:
        return
    fixed = asm_path + ".fixed"
    r2 = subprocess.run([sys.executable, tool, asm_path, "--fix", fixed], capture_output=True, text=True)
    if r2.returncode != 0:
        raise ExecSpillError(f"{what}: VGPR spill code before the exec restore of a join block, not repairable in the assembly "
                             f"(python tools/check_exec_spill.py; perturb the source or build the unit with -mllvm -amdgpu-spill-sgpr-to-vgpr=false):\n" + r2.stdout[-3000:])
    print(f"nuts_rs_amd.build: {what}: repaired in the assembly:\n" + r.stdout[-1500:], file=sys.stderr)
    os.replace(fixed, asm_path)
    # replay the pipeline from the device assembler on (the commands hipcc itself ran: `hipcc -###` with the same arguments)
    plan = subprocess.run(cmd + ["-###"], capture_output=True, text=True).stderr.split("\n")
    steps = [shlex.split(l) for l in plan if l.lstrip().startswith('"')]
    first = next((k for k, c in enumerate(steps) if "-cc1as" in c and "amdgcn-amd-amdhsa" in c), None)
    if first is None:
        raise ExecSpillError(f"{what}: cannot replay the compiler pipeline on the repaired assembly (no device assembler step in `hipcc -###`)")
    for c in steps[first:]:
        if "-E" in c and "-cc1" in c:          # (the host preprocessor step: its output is unchanged)
            continue
        subprocess.check_call(c)
    r3 = subprocess.run([sys.executable, tool, asm_path], capture_output=True, text=True)
    if r3.returncode != 0:
        raise ExecSpillError(f"{what}: the repaired assembly still has a finding:\n" + r3.stdout[-2000:])


def _compile_scanned(hipcc, flags, src, out, what, link=False, allow_calls=True):
    """Compile `src` (one hipcc run: -save-temps keeps the device assembly the object is assembled from), scan it, drop the temporaries."""
    import shutil
    import tempfile
    tmp = tempfile.mkdtemp(prefix="nm_tmp_", dir=os.path.dirname(out) or ".")
    try:
        # (-save-temps=obj writes next to the OUTPUT: compile into the temporary directory, then move the product out)
        prod = os.path.join(tmp, os.path.basename(out))
        cmd = [hipcc] + flags + ["-save-temps=obj"] + ([] if link else ["-c"]) + [src, "-o", prod]
        subprocess.check_call(cmd)
        asms = [os.path.join(tmp, f) for f in os.listdir(tmp) if f.endswith(".s") and "amdgcn" in f]
        if not asms and not src.endswith(".cpp"):
            raise RuntimeError(f"{what}: no device assembly among the compiler's temporaries (-save-temps=obj)")
        for a in asms:
            _repair_exec_spills(a, cmd, what)
            _scan_store_hazards(a, what, allow_calls)
        os.replace(prod, out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def build(force=False, verbose=False, extra_flags=(), scan=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]

    def compile_unit(u):
        src, obj = _src(u), _obj(u)
        if force or _unit_stale(u):
            variant = u.partition("@")[2]
            flags = FLAGS + VARIANT_FLAGS[variant] + list(extra_flags) + (["-Rpass-analysis=kernel-resource-usage"] if verbose else [])
            if verbose:
                print(" ".join([hipcc] + flags + ["-c", src, "-o", obj]), file=sys.stderr)
            if scan and src.endswith(".hip"):
                _compile_scanned(hipcc, flags, src, obj, os.path.basename(src) + ("@" + variant if variant else ""), allow_calls=variant not in NO_CALL_VARIANTS)
            else:
                subprocess.check_call([hipcc] + flags + ["-c", src, "-o", obj])
        return obj

    with ThreadPoolExecutor(max_workers=min(len(UNITS), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_unit, UNITS))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB])
    selftest_if_gpu()
    return LIB


def selftest_if_gpu():
    """The known-answer self-tests (nuts_rs_amd/selftest.py) on the library just built, when this machine has a GPU — in a fresh process (the
    library of the building process may already be mapped): the two small runs on the wave / group / lane kernels, then EVERY instantiation of
    the one-chain-per-block kernels against its own known answer (349 runs, ~1 min).  Without a GPU (the cross-compiling
    build box) nothing runs and a line says so: smoke() and the first engine of each instantiation run them on the GPU box."""
    probe = ("import sys, torch\n"
             "sys.exit(0 if torch.cuda.is_available() else 3)\n")
    try:
        if subprocess.run([sys.executable, "-c", probe], capture_output=True, timeout=300).returncode != 0:
            print("nuts_rs_amd.build: no GPU here (or no torch): the known-answer self-tests of the built library were NOT run", file=sys.stderr)
            return False
    except Exception as e:          # noqa: BLE001
        print(f"nuts_rs_amd.build: GPU probe failed ({e!r}): the known-answer self-tests were NOT run", file=sys.stderr)
        return False
    code = ("import nuts_rs_amd.selftest as s; print(s.run(), 'self-test runs ok'); "
            "print(s.run_all(), 'kernel instantiations reproduce their known answers')")
    try:
        r = subprocess.run([sys.executable, "-c", code], cwd=os.path.join(HERE, ".."), capture_output=True, text=True, timeout=1800,
                           env=dict(os.environ, NUTS_AMD_SELFTEST="0"))       # (run_all checks everything itself: no first-use checks inside it)
    except subprocess.TimeoutExpired:
        raise RuntimeError("the library was linked, but its known-answer self-test did not finish within 30 minutes on this GPU "
                           "(a hung kernel?): do not use this build") from None
    if r.returncode != 0:
        raise RuntimeError("the library just built FAILED its known-answer self-test on this GPU:\n" + r.stdout[-3000:] + r.stderr[-4000:])
    return True


def pick_tiling(dim, dims_per_lane=0, waves_per_chain=0):
    """(doubles per lane, waves per chain) the engine uses for `dim` — the same table as nm_pick_tiling."""
    for dpl, w in ((2, 1), (4, 1), (8, 1), (16, 1), (8, 2), (16, 2), (4, 4), (16, 4)):
        if dims_per_lane and dpl != dims_per_lane:
            continue
        if waves_per_chain and w != waves_per_chain:
            continue
        if dpl * 64 * w >= dim:
            return dpl, w
    raise ValueError(f"no tiling for dim {dim}")


def build_density_module(header, struct_name, dim, out, dims_per_lane=0, waves_per_chain=0, extra_flags=(), group_struct=None, lane_struct=None,
                         variants=()):
    """Compile a user density (a functor `struct_name` defined in `header`, see include/nuts_amd.h "User densities")
    with the engine's kernels into the module `out` for the tiling of `dim`.  Cross-compiles without a GPU (~20 s).
    variants: "low_rank" adds the kernels LowRankNutsSettings needs, "kinetic" those of the non-Euclidean trajectory kinds / MCLMC
    (each about doubles the build time)."""
    if dim > 4096:            # several blocks per chain: the cluster-mode kernels on the (16, 4) tiling; the density brings init_slice
        dpl, w = 16, 4
        extra_flags = list(extra_flags) + ["-DNM_CLUSTER_MODE=1"]
    else:
        dpl, w = pick_tiling(dim, dims_per_lane, waves_per_chain)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    header = os.path.abspath(header)
    if group_struct:          # the density's group form: several chains per wavefront for dim <= 64
        if dim > 64:
            raise ValueError("group forms exist for dim <= 64")
        extra_flags = list(extra_flags) + [f"-DNM_MODULE_GROUP_DENSITY={group_struct}", f"-DNM_MODULE_GS={8 if dim <= 16 else 16 if dim <= 32 else 32}"]
    if lane_struct:           # the density's lane form: one chain per lane for dim <= 10 (`template <int NP> struct ...`; the 8-pair kernel for dim 11 .. 16 was removed in round 5)
        if dim > 10:
            raise ValueError("lane forms exist for dim <= 10")
        extra_flags = list(extra_flags) + [f"-DNM_MODULE_LANE_DENSITY={lane_struct}"]
    # The special functions (nm::dexp / dlog / dlog1p, merge_math, ...) are INLINED in a module of a several-chains-per-wavefront form (a user functor
    # must not reach them through a call there: DESIGN §22, tools/check_divergent_calls.py) and, since round 5, in a module of a small tiling
    # (<= 4 doubles per lane) exactly as in the library's own units of those tilings: §22's fourth incident includes a dim-40 module whose
    # out-of-line build reported a wrong energy statistic while the inlined build of the same sources is bit-exact
    # (profiles/r05z_module_variants.txt), and a module is compiled on the USER's machine, where the engine's parity suite does not run.
    # Round 6: the defect behind those incidents is known (a VGPR spill placed above a join block's exec restore) and EVERY module
    # build — whatever its tiling, with or without calls — is scanned for it and repaired in the assembly (_repair_exec_spills); the calls
    # only made it likelier (SGPR spills around each call site).  The inlining rule stays as it was.
    no_calls = bool(group_struct or lane_struct) or dpl <= 4
    if no_calls:
        extra_flags = list(extra_flags) + ["-DNM_DETMATH_INLINE=1"]
    vbits = (1 if "low_rank" in variants else 0) | (2 if "kinetic" in variants else 0)
    if vbits:
        if dim > 4096:
            raise ValueError("the low-rank / kinetic variants exist for dim <= 4096")
        extra_flags = list(extra_flags) + [f"-DNM_MODULE_VARIANTS={vbits}"]
    extra_flags = list(extra_flags) + os.environ.get("NM_MODULE_EXTRA_FLAGS", "").split()       # (tuning / bisecting builds of a module)
    flags = FLAGS + list(extra_flags) + [
        "-shared", f"-DNM_MODULE_DENSITY={struct_name}", f'-DNM_MODULE_HEADER="{header}"', f"-DNM_MODULE_DPL={dpl}",
        f"-DNM_MODULE_W={w}", "-I", CSRC, "-I", os.path.join(HERE, "..", "include")]
    # (the user's code is compiled into the same register-capped kernels: its assembly goes through the same scan as the engine's own)
    _compile_scanned(hipcc, flags, os.path.join(CSRC, "density_module.hip"), os.path.abspath(out), f"density module {struct_name}", link=True, allow_calls=not no_calls)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
