#!/bin/bash
# Headline A/B on one box: LIBS="product xyz" (names after libgmr1_hip_), alternating, three rounds; kernel time.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-headline_ab}.txt
: > $OUT
lib_path() { if [ $1 = product ]; then echo $GRAFT_REPO_ROOT/osmo-gmr_amd/libgmr1_hip.so; else echo $GRAFT_REPO_ROOT/osmo-gmr_amd/libgmr1_hip_$1.so; fi; }
for rep in 1 2 3; do for lib in ${LIBS:-product}; do
	GMR1_HIP_LIBRARY=$(lib_path $lib) timeout -k 10 300 python3 bench.py --no-cpu --no-extras --steps 100 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0])
print('$lib il %.4f ms %.1f %%  step %.4f  checks %s' % (d['roofline']['kernel_ms'], 100 * d['roofline']['frac'], d['ms_per_step'], d.get('checks')))" >> $OUT || { echo "$lib failed" >> $OUT; cat $OUT; exit 1; }
done; done
cat $OUT
