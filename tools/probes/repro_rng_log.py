"""Round 5: the random-stream position of chain 1 at the points that move it, beside the draw-0 comparison with the oracle (fourth
incident; output: profiles/r05dbg_rng_position_log_failing_and_passing_builds.txt).  Needs a build of the unit with a log macro that is NOT in the
tree (it was applied to a worktree of 8e9c172):

    #define NM_DBG(C, tag, val) { if (blockIdx.x == 1 && threadIdx.x == 0) { const unsigned long long i_ = (C).P.prof[31]; (C).P.prof[31] = i_ + 1ull;
        if (i_ < 31ull) (C).P.prof[i_] = ((unsigned long long)(tag) << 56) | ((unsigned long long)(val) & 0xffffffffffffffull); } }

placed: before / after `sample_velocity` in nuts_transition (tags 01 / 02, value rng.pos), after the direction draw (10), after `C.rng.pos += mf & 2`
in merge_weights (0x20 | flags | is_main << 3; with 60 / 61 / 62 = a_log_size, b_log_size, total >> 8), after `C.rng.pos += ...` in resolve_chunk
(40: words << 40 | pos; 63 = the chunk's log size >> 8), at the top of update_stepsize (50).  `nm_debug_read_prof` returns the 32 words."""
import ctypes as C, os, sys
import numpy as np
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nuts_rs_amd as N
from nuts_rs_amd import _lib
from oracle import oracle as O
from helpers import oracle_settings
dim, md, n, draws = 130, int(sys.argv[1]) if len(sys.argv) > 1 else 2, 3, 2
s = N.LowRankNutsSettings(num_chains=n, seed=1234, num_tune=100, maxdepth=md)
logp = N.LogpSpec.iid_normal(dim, 0.3)
x0 = O.init_positions_uniform(s.seed, 0, n, dim)
b = N.ChainBatch(s, logp, n)
b.set_position(x0)
b.set_lowrank_estimator_place("device")
L = _lib.load()
buf = (C.c_ulonglong * 32)()
L.nm_debug_read_prof(b._h, buf)          # clear what set_position logged
pos, st = b.draw_many(1)
L.nm_debug_read_prof(b._h, buf)
log0 = list(buf)
pos2, st2 = b.draw_many(1)
tpc = b.threads_per_chain()
b.close()
est = dict(estimator=C.cast(L.nm_lowrank_block_twin, O.ESTIMATOR_FN))
pos_o, st_o, _, _ = O.run(oracle_settings(O, s), logp.kind, dim, logp.params, O.gpu_cfg(tpc), n, x0, draws, n_threads=1, **est)
print("lib", os.path.basename(os.environ.get("NUTS_AMD_LIB", "default")), "maxdepth", md)
for c in range(n):
    print(f" draw 0 chain {c}: dpos {np.abs(pos[0, c] - pos_o[0, c]).max():.2e} step {st['step_size'][0, c]!r} / {st_o['step_size'][0, c]!r}   draw 1 energy {st2['energy'][0, c]!r} / {st_o['energy'][1, c]!r}")
print(" chain 1 log (tag: 01/02 momentum before/after, 10 direction, 2x merge_weights [bit0 take, bit1 drew a word, bit2 fatal, bit3 main tree], 40 chunk [words << 40], 50 step-size update):")
for v in log0[:31]:
    if v:
        tag = v >> 56
        if tag >= 0x60:
            import struct
            print(f"   tag {tag:02x}  value ~ {struct.unpack('<d', struct.pack('<Q', (v & 0xffffffffffffff) << 8))[0]!r}  bits>>8 {v & 0xffffffffffffff:014x}")
        else:
            print(f"   tag {tag:02x}  pos {v & 0xffffffffff}  extra {(v >> 40) & 0xffff}")
print(" entries", log0[31])
