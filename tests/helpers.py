"""Shared helpers for the parity tests: run the same configuration on the HIP engine and on the CPU oracle."""
import numpy as np

import nuts_rs_amd as N

STAT_FIELDS_EXACT = ["draw", "chain", "depth", "maxdepth_reached", "diverging", "tuning", "n_steps",
                     "index_in_trajectory", "transformation_index", "chain_status", "transformation_update_id"]
STAT_FIELDS_FLOAT = ["step_size", "step_size_bar", "mean_tree_accept", "mean_tree_accept_sym", "max_energy_error",
                     "logp", "energy", "energy_error", "fisher_distance"]


def oracle_settings(O, settings: N.DiagNutsSettings):
    """DiagNutsSettings -> the oracle's Settings struct (same field names)."""
    c = settings.to_c()
    s = O.Settings()
    for name, _ in O.Settings._fields_:
        setattr(s, name, getattr(c, name))
    return s


def run_engine(settings, logp, n_chains, x0, n_draws, chain_id_offset=0, dims_per_lane=0, waves_per_chain=0,
               lane_groups=0, grid_blocks=0, splits=(), lane_chains=0):
    """`splits`: draw counts at which the run is cut into separate launches; the pieces are concatenated."""
    b = N.ChainBatch(settings, logp, n_chains, chain_id_offset=chain_id_offset, dims_per_lane=dims_per_lane,
                     waves_per_chain=waves_per_chain, lane_groups=lane_groups, grid_blocks=grid_blocks, lane_chains=lane_chains)
    status = b.set_position(x0, raise_on_error=False)
    pos, st = None, None
    if (status == 0).all():
        cuts = [0] + [c for c in splits if 0 < c < n_draws] + [n_draws]
        parts = [b.draw_many(hi - lo) for lo, hi in zip(cuts[:-1], cuts[1:])]
        pos, st = np.concatenate([p for p, _ in parts]), np.concatenate([q for _, q in parts])
    extra = dict(status=status, threads_per_chain=b.threads_per_chain(), dims_per_lane=b.dims_per_lane(),
                 group_launches=b.group_launches(), lane_launches=b.lane_launches())
    if pos is not None:
        sd, mu = b.mass_matrix()
        extra.update(stds=sd, mean=mu, step_sizes=b.step_sizes(), x=b.positions(), gx=b.gradients(),
                     counters=b.counters())
    b.close()
    return pos, st, extra


def run_oracle(O, settings, logp, n_chains, x0, n_draws, chain_id_offset=0, gpu_threads=64, cfg=None, n_threads=8):
    cfg = cfg or O.gpu_cfg(gpu_threads)
    return O.run(oracle_settings(O, settings), logp.kind, logp.dim, logp.params, cfg, n_chains, x0, n_draws,
                 chain_offset=chain_id_offset, n_threads=n_threads)


def assert_bit_exact(pos_g, st_g, pos_o, st_o):
    """Draw-for-draw, bit-for-bit: positions and every statistic."""
    for f in STAT_FIELDS_EXACT:
        bad = np.argwhere(st_g[f] != st_o[f])
        assert bad.size == 0, f"stat {f} differs first at (draw, chain) = {bad[0]}: gpu {st_g[f][tuple(bad[0])]} oracle {st_o[f][tuple(bad[0])]}"
    for f in STAT_FIELDS_FLOAT + ["divergence_energy_error"]:
        a, b = st_g[f].view(np.uint64), st_o[f].view(np.uint64)
        both_nan = np.isnan(st_g[f]) & np.isnan(st_o[f])
        bad = np.argwhere((a != b) & ~both_nan)
        assert bad.size == 0, f"stat {f} differs first at (draw, chain) = {bad[0]}: gpu {st_g[f][tuple(bad[0])]!r} oracle {st_o[f][tuple(bad[0])]!r}"
    bad = np.argwhere(pos_g.view(np.uint64) != pos_o.view(np.uint64))
    assert bad.size == 0, f"positions differ first at (draw, chain, dim) = {bad[0]}"


def assert_vectors_bit_exact(vec_g, vec_o):
    """Vector-valued statistics: same bits, and the same rows left unwritten (NaN)."""
    assert set(vec_g) <= set(vec_o)
    for k, g in vec_g.items():
        o = vec_o[k]
        both_nan = np.isnan(g) & np.isnan(o)
        bad = np.argwhere((g.view(np.uint64) != o.view(np.uint64)) & ~both_nan)
        assert bad.size == 0, f"{k} differs first at (draw, chain, dim) = {bad[0]}: gpu {g[tuple(bad[0])]!r} oracle {o[tuple(bad[0])]!r}"
