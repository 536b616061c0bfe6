# Fabric-side traffic of the configs[4]-from-samples step (bench.py --workload nt3, 1 M bursts): rocprofv3 --pmc FETCH_SIZE /
# WRITE_SIZE / TCC hit-miss in separate passes with --kernel-trace only, FETCH_SIZE doubled as MI355X_MICROARCH.md
# prescribes for gfx950.  Writes gpurun_out/hbm_traffic_nt3.json (copy to profiles/): per kernel and summed over the step,
# with the hash of the kernel sources it was taken on (bench.py ignores the file when the hash differs).
#   bash tools/traffic_nt3.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/tnt3_${tag}_$n -- python3 bench.py --workload nt3 --no-cpu --preroll-s 0 --steps 4 --warmup 1 > gpurun_out/tnt3_${tag}_$n.log 2>&1
done
python3 - $tag <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum"):
    for f in glob.glob(f"gpurun_out/tnt3_{tag}_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gmr1::" in k and "k_coef0" not in k and "k_to_planar" not in k:
                acc[k.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
sys.path.insert(0, ".")
import bench
kernels, total = {}, 0
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    hit, miss = m.get("TCC_HIT_sum", 0), m.get("TCC_MISS_sum", 0)
    b = int(2 * m.get("FETCH_SIZE", 0) * 1024 + m.get("WRITE_SIZE", 0) * 1024)
    kernels[k] = {"fetch_size_raw_kb": m.get("FETCH_SIZE", 0), "write_size_raw_kb": m.get("WRITE_SIZE", 0), "bytes_per_launch": b,
                  "tcc_hit_rate": hit / (hit + miss) if hit + miss else None, "launches_averaged": len(c.get("FETCH_SIZE", []))}
    total += b
out = {"_comment": "Fabric-side bytes of one step of bench.py --workload nt3 (1 000 000 NT3 bursts: 900 000 speech through k_rx4g_tch3, "
                   "100 000 FACCH3 through k_rx4g<8,4,FAC> + k_facch3): rocprofv3 --pmc, separate passes, FETCH_SIZE doubled "
                   "(tools/traffic_nt3.sh).  Algorithmic bytes of the step: 3 829 850 000.",
       "tag": tag, "kernels": kernels, "step_bytes_1M": total, "kernel_sources_sha256": bench.kernel_sources_hash()}
json.dump(out, open("gpurun_out/hbm_traffic_nt3.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf gpurun_out/tnt3_${tag}_*
