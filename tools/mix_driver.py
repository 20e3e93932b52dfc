"""One workload for tools/pmc_mix.py: warm-up launch, then ONE timed launch whose counters are the last dispatch of the kernel.
  python tools/mix_driver.py k2|k2one|k3|k3one|k4g|k4one [draws]
Prints a JSON line with the leapfrogs of the timed launch."""
import json, os, sys
import torch  # noqa: F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nuts_rs_amd as N  # noqa: E402
key = sys.argv[1]
draws = int(sys.argv[2]) if len(sys.argv) > 2 else 100
CFG = {"k2": (lambda: N.LogpSpec.iid_normal(1024, 3.0), 4096, 100), "k2one": (lambda: N.LogpSpec.iid_normal(1024, 3.0), 1, 100),
       "k3": (lambda: N.LogpSpec.funnel(101), 8192, 400), "k3one": (lambda: N.LogpSpec.funnel(101), 1, 400),
       "k4g": (N.LogpSpec.eight_schools, 8192, 400), "k4one": (N.LogpSpec.eight_schools, 8, 400)}
CFG["k3deep"] = (lambda: N.LogpSpec.funnel(101), 1, 1)          # tools/leaf_latency.py's workload: one funnel chain, every tree to depth 8
CFG["k3deep1024"] = (lambda: N.LogpSpec.funnel(101), 1024, 1)
mk, chains, tune = CFG[key]
logp = mk()
s = N.DiagNutsSettings(num_chains=chains, seed=20260928 if not key.startswith("k3deep") else 11, num_tune=tune, num_draws=draws, **({"maxdepth": 8} if key.startswith("k3deep") else {}))
if key.startswith("k3deep"):
    st = s.adapt_options.step_size_settings
    st.method, st.fixed_step_size, st.jitter = N.sampler.STEP_FIXED, 0.002, None
b = N.ChainBatch(s, logp, chains, lane_groups=2 if key == "k4one" else 0)
b.set_position(b.init_positions_uniform())
b.draw_device(tune)
b.reset_counters()
b.draw_device(draws)
c = b.counters()
print(json.dumps({"mix_driver": key, "chains": chains, "dim": logp.dim, "draws": draws, "leapfrogs": c["total_leapfrogs"], "kernel_ms": c["kernel_ms"],
                  "threads_per_chain": b.threads_per_chain(), "group_launches": b.group_launches(), "lane_launches": b.lane_launches()}), flush=True)
