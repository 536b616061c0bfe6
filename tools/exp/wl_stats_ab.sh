#!/bin/bash
# rocprofv3 --stats of one bench workload per library: WL=fcch LIBS="product sw1" TAG=x tools/exp/wl_stats_ab.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG:-wl_stats}.txt
: > $OUT
lib_path() { if [ $1 = product ]; then echo $R/osmo-gmr_amd/libgmr1_hip.so; else echo $R/osmo-gmr_amd/libgmr1_hip_$1.so; fi; }
for lib in ${LIBS:-product}; do
	export GMR1_HIP_LIBRARY=$(lib_path $lib)
	mkdir -p $R/gpurun_out/wl_prof
	timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/wl_prof -o $lib -- python3 $R/bench.py --workload ${WL:-fcch} --steps 20 --warmup 3 --no-cpu --no-extras > $R/gpurun_out/wl_prof/$lib.json 2>/dev/null || { echo "rocprof $lib failed" >> $OUT; cat $OUT; exit 1; }
	echo "== $lib" >> $OUT; cut -d, -f1-4 $R/gpurun_out/wl_prof/${lib}_kernel_stats.csv | head -5 | cut -c1-150 >> $OUT
done
cat $OUT
