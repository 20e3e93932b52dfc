"""Stream discipline of the native library (DESIGN "Stream discipline").

Every handle of the library runs its work on its own hipStreamNonBlocking stream.  The null stream is NOT ordered against
such a stream, so a null-stream fill / copy / kernel on memory the library owns may overtake, or be overtaken by, the
handle's kernels (round 2's red test was exactly that: `hipMemset` in nm_vec_new against the first kernel's store).  The
rule: no blocking `hipMemset(` / `hipMemcpy(` / `hipMemcpy2D(` and no null-stream kernel launch in the product sources; copies
go through copy_on / copy2d_on (enqueue on the owning stream, wait for that stream).  This test keeps the rule by reading
the sources; it needs no GPU."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nuts_rs_amd", "csrc")


def _code_lines(path):
    for no, line in enumerate(open(path), 1):
        code = line.split("//", 1)[0]
        if code.strip():
            yield no, code


def _sources():
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".hpp", ".cpp")):
            yield os.path.join(CSRC, f)
    inc = os.path.join(ROOT, "include")
    for f in sorted(os.listdir(inc)):
        yield os.path.join(inc, f)


def test_no_null_stream_memory_operations():
    forbidden = re.compile(r"\bhip(Memset|MemsetD8|MemsetD16|MemsetD32|Memcpy|Memcpy2D|MemcpyHtoD|MemcpyDtoH|MemcpyDtoD)\s*\(")
    hits = [f"{os.path.relpath(p, ROOT)}:{no}: {code.strip()}" for p in _sources() for no, code in _code_lines(p) if forbidden.search(code)]
    assert not hits, "blocking / null-stream memory operations in the product sources:\n" + "\n".join(hits)


def test_no_null_stream_kernel_launches():
    # hipLaunchKernelGGL(kernel, grid, block, lds, STREAM, ...): the stream argument must not be a literal null
    launch = re.compile(r"hipLaunchKernelGGL\s*\((.*)")
    hits = []
    for p in _sources():
        for no, code in _code_lines(p):
            m = launch.search(code)
            if not m:
                continue
            args, depth, cur = [], 0, ""
            for ch in m.group(1):
                if ch in "(<[{":
                    depth += 1
                elif ch in ")>]}":
                    depth -= 1
                    if depth < 0:
                        break
                if ch == "," and depth == 0:
                    args.append(cur.strip()); cur = ""
                else:
                    cur += ch
            args.append(cur.strip())
            if len(args) >= 5 and args[4] in ("0", "nullptr", "NULL", "hipStreamDefault"):
                hits.append(f"{os.path.relpath(p, ROOT)}:{no}: {code.strip()}")
    assert not hits, "kernel launches on the null stream:\n" + "\n".join(hits)


def test_handles_create_nonblocking_streams_only():
    hits = [f"{os.path.relpath(p, ROOT)}:{no}" for p in _sources() for no, code in _code_lines(p)
            if re.search(r"\bhipStreamCreate\s*\(", code)]
    assert not hits, "plain hipStreamCreate (a blocking stream synchronises implicitly with the null stream; the library relies on none of that): " + ", ".join(hits)
