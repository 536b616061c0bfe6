# Fabric-side traffic of the NT3 workload's kernels (rocprofv3 --pmc in separate passes, --kernel-trace only; FETCH_SIZE
# doubled as MI355X_MICROARCH.md prescribes for gfx950):   bash tools/traffic_nt3.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/tnt3_${tag}_$n -- python3 bench.py --workload nt3 --no-cpu --preroll-s 0 --steps 4 --warmup 1 > gpurun_out/tnt3_${tag}_$n.log 2>&1
done
python3 - $tag <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum"):
    for f in glob.glob(f"gpurun_out/tnt3_{tag}_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gmr1::" in k:
                acc[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    hit, miss = m.get("TCC_HIT_sum", 0), m.get("TCC_MISS_sum", 0)
    print(f"{k}: fetch {2 * m.get('FETCH_SIZE', 0) / 1e6 * 1.024:.1f} MB (raw KB x 2), write {m.get('WRITE_SIZE', 0) / 1e6 * 1.024:.1f} MB, "
          f"L2 hit rate {hit / (hit + miss) if hit + miss else 0:.3f}  (n={len(c.get('FETCH_SIZE', []))})")
PY
rm -rf gpurun_out/tnt3_${tag}_*
