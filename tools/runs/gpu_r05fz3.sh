#!/bin/bash
# round 5: more randomised parity on the binary of record r05fin6 (seeds 711 - 716, 120 cases each; scale seeds 721 - 723, 25 each)
export TMPDIR=/tmp; O=gpurun_out/r05fz3; mkdir -p $O
for S in 811 812 813 814 815 816; do timeout 900 python tools/fuzz_parity.py --cases 120 --seed $S > $O/fuzz_$S.txt 2>&1; echo "seed $S: $(tail -1 $O/fuzz_$S.txt)" >> $O/summary.txt; done
for S in 821 822 823; do timeout 900 python tools/fuzz_parity.py --cases 25 --scale --seed $S > $O/fuzz_scale_$S.txt 2>&1; echo "scale seed $S: $(tail -1 $O/fuzz_scale_$S.txt)" >> $O/summary.txt; done
grep -h MISMATCH $O/fuzz_*.txt >> $O/summary.txt
cat $O/summary.txt
