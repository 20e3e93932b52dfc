#!/bin/bash
# round 6: the shared gradient tile for densities that are not element-wise (NM_GTILE_ALL) on the one-wavefront 8- / 16-doubles tilings: parity, then rates
export TMPDIR=/tmp; O=$PWD/gpurun_out/r06y; mkdir -p $O; export NUTS_AMD_SELFTEST=0
export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_ga.so
timeout 900 python -m pytest tests/test_gpu_every_instantiation.py tests/test_gpu_parity.py -x -q -m gpu -k "funnel or mvn" 2>&1 | tail -3 > $O/parity.txt; cat $O/parity.txt
cat > /tmp/rate.py <<'PY'
import sys, time, numpy as np
import nuts_rs_amd as N
def run(name, logp, chains, tune, draws, **eng):
    s = N.DiagNutsSettings(num_chains=chains, seed=77, num_tune=tune, num_draws=draws)
    b = N.ChainBatch(s, logp, chains, **eng)
    b.set_position(b.init_positions_uniform())
    b.draw_device(tune); b.reset_counters()
    t = time.time(); b.draw_device(draws); dt = time.time() - t
    c = b.counters(); print(name, "tiling", b.dims_per_lane(), b.threads_per_chain() // 64, "leapfrogs/s %.4g" % (c["total_leapfrogs"] / dt), "kernel ms %.1f" % c["kernel_ms"], "steps/draw %.1f" % (c["total_leapfrogs"] / draws / chains)); b.close()
r = np.random.default_rng(5)
run("funnel dim 1000", N.LogpSpec.funnel(1000), 4096, 300, 100)
run("funnel dim 500 ", N.LogpSpec.funnel(500), 4096, 300, 100)
a = r.normal(size=(700, 8)); p = a @ a.T / 8 + np.eye(700)
run("mvn dim 700    ", N.LogpSpec.mvn_precision((p + p.T) / 2), 2048, 200, 50, chain_tiles=1)
a = r.normal(size=(400, 8)); p = a @ a.T / 8 + np.eye(400)
run("mvn dim 400    ", N.LogpSpec.mvn_precision((p + p.T) / 2), 2048, 200, 50, chain_tiles=1)
PY
for rep in 1 2; do for L in "" _ga; do
  export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd$L.so
  echo "== lib $L" >> $O/rates.txt
  PYTHONPATH=$PWD timeout 600 python /tmp/rate.py >> $O/rates.txt 2>&1
done; done
cat $O/rates.txt
