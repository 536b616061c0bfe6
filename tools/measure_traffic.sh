# HBM-side traffic of the headline kernel, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in separate
# --pmc passes (with --kernel-trace only), FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B; calibrated on
# k_rx, which reads every byte once: raw 386.6 MB for 789 MB).  Writes profiles/hbm_traffic.json with the hash of the
# kernel sources it was taken on; bench.py ignores the file when the hash differs.   bash tools/measure_traffic.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/traffic_${tag}_$n -- python3 bench.py --no-cpu --no-extras --preroll-s 0.05 --steps 10 > gpurun_out/traffic_${tag}_$n.log 2>&1
  # the same three passes with the opt-in polyphase-planar layout as the timed step (bench.py --layout planar)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/traffic_${tag}_pl_$n -- python3 bench.py --no-cpu --layout planar --preroll-s 0.05 --steps 10 > gpurun_out/traffic_${tag}_pl_$n.log 2>&1
done
python3 - $tag <<'PY'
import csv, glob, hashlib, json, os, sys
tag = sys.argv[1]
def mean(counter, d):
    v = []
    for f in glob.glob(f"gpurun_out/traffic_{tag}_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_rx4<" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                v.append(float(r["Counter_Value"]))
    return sum(v) / len(v), len(v)
fetch, n = mean("FETCH_SIZE", "FETCH_SIZE")
write, _ = mean("WRITE_SIZE", "WRITE_SIZE")
hit, _ = mean("TCC_HIT_sum", "TCC_HIT_sum")
miss, _ = mean("TCC_MISS_sum", "TCC_HIT_sum")
pfetch, pn = mean("FETCH_SIZE", "pl_FETCH_SIZE")
pwrite, _ = mean("WRITE_SIZE", "pl_WRITE_SIZE")
phit, _ = mean("TCC_HIT_sum", "pl_TCC_HIT_sum")
pmiss, _ = mean("TCC_MISS_sum", "pl_TCC_HIT_sum")
sys.path.insert(0, ".")
import bench
out = {"_comment": "Fabric-side bytes per launch of k_rx4<16,4> over 100000 bursts: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate "
                   "passes (tools/measure_traffic.sh), FETCH_SIZE doubled per MI355X_MICROARCH.md. The counters sit above the Infinity "
                   "Cache: its hits are included. Since round 6 pass 2's kept samples come out of the window registers at a speculated pick; "
                   "only the bursts whose pick is another (about 15 %) read their window a second time.",
       "tag": tag, "launches_averaged": n, "fetch_size_raw_kb": fetch, "write_size_raw_kb": write,
       "tcc_hit_rate": hit / (hit + miss) if hit + miss else None,
       "k_rx_bytes_per_launch_100k": int(2 * fetch * 1024 + write * 1024),
       "planar": {"_comment": "the same counters for k_rx4<16,4,*,false,true> (gmr1_hip_rx_bcch_ccch_batch_planar_dev, bench.py --layout planar)",
                  "launches_averaged": pn, "fetch_size_raw_kb": pfetch, "write_size_raw_kb": pwrite,
                  "tcc_hit_rate": phit / (phit + pmiss) if phit + pmiss else None},
       "k_rx_planar_bytes_per_launch_100k": int(2 * pfetch * 1024 + pwrite * 1024),
       "kernel_sources_sha256": bench.kernel_sources_hash()}
json.dump(out, open("gpurun_out/hbm_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
