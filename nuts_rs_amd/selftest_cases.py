"""The known-answer cases of every kernel instantiation of the one-chain-per-block family.

One definition, three users: `tests/test_gpu_every_instantiation.py` (engine against the oracle, bit for bit), `tools/gen_selftest_golden.py`
(the oracle's answers as DATA: nuts_rs_amd/selftest_instantiations.json) and `nuts_rs_amd.selftest` (the built library against that data — in
build() on a GPU box, and for ONE instantiation the first time a process creates an engine that runs it).

    density  x  {DiagNutsSettings, LowRankNutsSettings frozen / adapting, ExactNormal, Microcanonical, MCLMC, LowRank MCLMC}
             x  the eight tilings (doubles per lane, wavefronts per chain)  x  both ends of the tiling's range of dims

The draw every one of these kernels restates: src/chain.rs:150-243; the settings families: src/sampler.rs:199-245, :266-384 (reference).
Nothing here touches the GPU or the oracle: it only says WHAT is run."""
import numpy as np

TILINGS = [(2, 1), (4, 1), (8, 1), (16, 1), (8, 2), (16, 2), (4, 4), (16, 4)]
FAMILIES = ["nuts", "lr_frozen", "lr_adapt", "exact", "micro", "mclmc", "lr_mclmc"]
DENSITIES = ("iid", "diag", "funnel", "mvn", "schools")
# nm_logp_spec.kind of the built-in densities (include/nuts_amd.h) -> the name used here
KIND_NAME = {0: "iid", 1: "diag", 2: "funnel", 3: "schools", 4: "mvn"}


def case_id(c):
    return f"{c['dens']}-{c['fam']}-{c['dpl']}x{c['w']}-dim{c['dim']}"


def cases(both_ends=True):
    """Every (density, family, tiling) of the one-chain-per-block kernels, at the top of the tiling's range of dims (the cases of round 5) and,
    with `both_ends`, at its bottom (the fourth incident of DESIGN §22 was found at dim 129 = the first dim of the (4,1) tiling)."""
    out = []
    lower = {}
    for (d, w) in TILINGS:
        cap = d * 64 * w
        lower[(d, w)] = max(c for c in [0] + [dd * 64 * ww for (dd, ww) in TILINGS] if c < cap) + 1
    for dens in DENSITIES:
        for (d, w) in TILINGS:
            cap = d * 64 * w
            if dens == "schools" and (d, w) != (2, 1):
                continue
            if dens == "mvn" and cap > 2048:
                continue
            for fam in FAMILIES:
                if fam in ("lr_adapt", "lr_mclmc") and cap > 256:   # the device estimator's range here (the oracle runs its twin); beyond: test_gpu_lowrank, test_gpu_mclmc
                    continue
                # full tiles on every second tiling for the densities whose kernels have a full-tile path, ragged tiles otherwise
                full = (TILINGS.index((d, w)) % 2 == 1) and dens in ("iid", "diag", "mvn")
                dim = 10 if dens == "schools" else (cap if full else cap - 3)
                if dens == "mvn":
                    dim = min(dim, 700)                      # a dense precision matrix: O(dim^2) per leapfrog in the oracle
                out.append(dict(dens=dens, fam=fam, dpl=d, w=w, dim=dim, end="top"))
                if both_ends and dens != "schools":
                    lo = max(lower[(d, w)], 2)               # (the funnel and MCLMC need dim >= 2)
                    if dens == "mvn":
                        lo = min(lo, 600)
                    if lo != dim and lo >= 2:
                        out.append(dict(dens=dens, fam=fam, dpl=d, w=w, dim=lo, end="bottom"))
    return out


def _dyadic(r, n, lo_exp=-2, hi_exp=2):
    """n positive doubles (1 + k/8) * 2^e: built from integers by exact operations only.  The inputs of a known-answer run must have the
    same BITS on every machine — no exp / log (numpy dispatches them to different SIMD implementations by CPU), no BLAS, no LAPACK (round 6:
    the first set of answers, whose precision matrices came from `a @ a.T`, failed on the GPU box for every mvn case of dim > 256)."""
    return np.ldexp(1.0 + r.integers(0, 8, n) / 8.0, r.integers(lo_exp, hi_exp + 1, n))


def make_logp(N, dens, dim, seed):
    r = np.random.default_rng(seed)
    if dens == "iid":
        return N.LogpSpec.iid_normal(dim, float(r.integers(-128, 129)) / 64.0)
    if dens == "diag":
        return N.LogpSpec.diag_normal(_dyadic(r, dim))
    if dens == "funnel":
        return N.LogpSpec.funnel(dim)
    if dens == "schools":
        return N.LogpSpec.eight_schools()
    a = r.integers(-3, 4, size=(dim, 8)).astype(np.int64)
    p = (a @ a.T).astype(np.float64) / 8.0 + np.eye(dim)       # integer products: exact; symmetric by construction
    return N.LogpSpec.mvn_precision(p)


def make_transform(dim, seed):
    """A frozen low-rank transformation (stds, mean, eigenvalues, orthonormal eigenvectors, gradient mean) of exactly representable numbers:
    the eigenvectors have entries +-1/2 on disjoint groups of four coordinates (unit coordinate vectors when dim < 20)."""
    r = np.random.default_rng(seed)
    if dim >= 20:
        rank = 5
        vecs = np.zeros((rank, dim))
        for k in range(rank):
            vecs[k, 4 * k:4 * k + 4] = 0.5 * (1 - 2 * r.integers(0, 2, 4))
    else:
        rank = min(dim, 5)
        vecs = np.eye(dim)[:rank].copy()
    return (_dyadic(r, dim, -1, 1), r.integers(-16, 17, dim) / 16.0, _dyadic(r, rank, -1, 2), np.ascontiguousarray(vecs), r.integers(-8, 9, dim) / 16.0)


def make_run(N, c):
    """-> dict(settings, logp, transform (None | 'adapt' | the five arrays of set_transform), draws, n_chains, engine keyword arguments)."""
    dens, fam, dpl, wpc, dim = c["dens"], c["fam"], c["dpl"], c["w"], c["dim"]
    n, seed = 3, 1000 + 17 * dpl + wpc + (0 if c.get("end", "top") == "top" else 5000)
    num_tune = 40 if dim > 600 else 70
    draws = num_tune + 10
    kw = dict(num_chains=n, seed=seed, num_tune=num_tune)
    transform = None
    if fam in ("mclmc", "lr_mclmc"):
        mk = N.DiagMclmcSettings if fam == "mclmc" else N.LowRankMclmcSettings
        if fam == "lr_mclmc":
            kw["num_tune"] = num_tune = 100
            draws = 110
            transform = "adapt"
        s = mk(step_size=0.4, momentum_decoherence_length=3.0, trajectory_kind=1, dynamic_step_size=True, subsample_frequency=0.5, **kw)
    elif fam in ("lr_frozen", "lr_adapt"):
        if fam == "lr_adapt":
            kw["num_tune"] = num_tune = 100
            draws = 110
        s = N.LowRankNutsSettings(freeze_transform=(fam == "lr_frozen"), maxdepth=6, **kw)
        if fam == "lr_adapt":
            s.adapt_options.mass_matrix_update_freq = 5
            transform = "adapt"
        else:
            transform = make_transform(dim, seed + 1)
    else:
        s = N.DiagNutsSettings(maxdepth=6, trajectory_kind={"nuts": 0, "exact": 1, "micro": 2}[fam], **kw)
    logp = make_logp(N, dens, dim, seed)
    eng = {} if dens == "schools" else dict(dims_per_lane=dpl, waves_per_chain=wpc)
    return dict(settings=s, logp=logp, transform=transform, draws=draws, n_chains=n,
                engine=dict(lane_groups=1, lane_chains=1, chain_tiles=1, **eng))     # the one-chain-per-block kernels, nothing else


def family_of(settings):
    """The settings families of `cases()` whose kernels an engine with these settings launches (the LrWrap / KinWrap / plain instantiation and its
    sampler).  `LowRankNutsSettings` / `LowRankMclmcSettings` are the Diag classes with `LowRankSettings` as mass_matrix_options (sampler.py)."""
    mo = getattr(getattr(settings, "adapt_options", None), "mass_matrix_options", None)
    low_rank = type(mo).__name__ == "LowRankSettings"
    if type(settings).__name__ == "DiagMclmcSettings":
        return ["lr_mclmc"] if low_rank else ["mclmc"]
    if low_rank:
        return ["lr_frozen", "lr_adapt"]
    tk = getattr(settings, "trajectory_kind", 0)
    kind = int(getattr(tk, "value", tk) or 0)
    return [{0: "nuts", 1: "exact", 2: "micro"}.get(kind, "nuts")]


def digest(pos, st):
    """What a run is compared by: SHA-256 over the bits of every position and of the statistics that are compared exactly in the parity tests."""
    import hashlib
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(pos, dtype=np.float64).tobytes())
    for f in ("depth", "n_steps", "diverging", "index_in_trajectory", "chain_status"):
        h.update(np.ascontiguousarray(st[f]).astype(np.int64).tobytes())
    for f in ("step_size", "energy", "logp", "mean_tree_accept", "energy_error"):
        a = np.ascontiguousarray(st[f], dtype=np.float64).copy()
        a[np.isnan(a)] = np.nan                   # one NaN
        h.update(a.tobytes())
    return h.hexdigest()
