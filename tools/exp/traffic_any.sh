#!/bin/bash
# Fabric-side traffic of the kernels of one bench workload: FETCH_SIZE / WRITE_SIZE / TCC hit rate in separate --pmc passes
# (--kernel-trace only), FETCH_SIZE doubled as for the headline.   WL=fcch KERNELS="k_fcch_sweep k_fcch_energy" TAG=x tools/exp/traffic_any.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${TAG:-any}
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/traffic_${tag}_$n -- python3 bench.py --workload ${WL:-fcch} --no-cpu --no-extras --preroll-s 0.05 --steps 5 --warmup 2 > gpurun_out/traffic_${tag}_$n.log 2>&1 || exit 1
done
python3 - $tag $KERNELS <<'PY' | tee gpurun_out/traffic_${tag}.txt
import csv, glob, sys
tag = sys.argv[1]
def mean(counter, d, kern):
    v = []
    for f in glob.glob(f"gpurun_out/traffic_{tag}_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"] and r["Counter_Name"] == counter:
                v.append(float(r["Counter_Value"]))
    return (sum(v) / len(v), len(v)) if v else (float("nan"), 0)
for kern in sys.argv[2:]:
    f, n = mean("FETCH_SIZE", "FETCH_SIZE", kern)
    w, _ = mean("WRITE_SIZE", "WRITE_SIZE", kern)
    h, _ = mean("TCC_HIT_sum", "TCC_HIT_sum", kern)
    m, _ = mean("TCC_MISS_sum", "TCC_HIT_sum", kern)
    print(f"{kern}: {n} launches; fetched {2 * f * 1024 / 1e6:.1f} MB (raw {f:.0f} KB x 2), written {w * 1024 / 1e6:.1f} MB, L2 hit rate {h / (h + m):.3f}")
PY
