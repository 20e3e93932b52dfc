#!/bin/bash
# round 3, first GPU pass: (1) the nm_vec_new race before / after, (2) the full suite on the new binary, core first,
# (3) single-chain leapfrog latency and the phase anatomy of a deep funnel tree (baseline for the K3 work)
export TMPDIR=/tmp
O=gpurun_out/r03a; mkdir -p $O
echo "== race probe, round-2 binary" | tee $O/race.log
NUTS_AMD_LIB=$PWD/tools/probes/_old/libnuts_amd_r02.so timeout 300 python tools/probes/vec_new_race.py 3000 4000 >> $O/race.log 2>&1
echo "== race probe, this build" >> $O/race.log
timeout 300 python tools/probes/vec_new_race.py 3000 4000 >> $O/race.log 2>&1
tail -4 $O/race.log
timeout 1700 python -m pytest tests -m gpu -x -q --durations=25 > $O/pytest.log 2>&1; tail -40 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for lp in funnel iid; do
  timeout 300 python tools/leaf_latency.py --logp $lp --dim 101 --maxdepth 8 --draws 20 --chains 1,256,1024,8192 >> $O/leaf_latency.jsonl 2>> $O/leaf.err
done
timeout 120 python tools/leaf_latency.py --logp iid --dim 1024 --maxdepth 8 --draws 10 --chains 1,1024,4096 >> $O/leaf_latency.jsonl 2>> $O/leaf.err
timeout 120 python tools/leaf_latency.py --logp schools --dim 10 --maxdepth 8 --draws 20 --chains 1,8192,65536 >> $O/leaf_latency.jsonl 2>> $O/leaf.err
cat $O/leaf_latency.jsonl; tail -3 $O/leaf.err
echo "== phases: funnel dim 101, 1 chain, fixed step, depth 8" > $O/phases.log
NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_prof.so PROF_LOGP=funnel PROF_FIXED_STEP=0.002 PROF_MAXDEPTH=8 timeout 300 python tools/prof_phases.py 1 101 1 20 >> $O/phases.log 2>&1
echo "== phases: funnel dim 101, 8192 chains, adaptive" >> $O/phases.log
NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_prof.so PROF_LOGP=funnel timeout 300 python tools/prof_phases.py 8192 101 400 100 >> $O/phases.log 2>&1
cat $O/phases.log
