#!/usr/bin/env python3
"""bench.py's native sharded leg (_sharded_native) on a world of ONE rank under backend nccl -- all a one-GPU box can
rehearse of it: the id broadcast, gmr1_hip_shard_create, two gmr1_hip_rx_run_sharded calls, the timing reduction."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist
import bench, workloads
from __graft_entry__ import load_package
pkg = load_package(); pkg.api.load(); pkg.api.init(0)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
A, sps, seconds, distinct = 6, 4, 4.0, 3
ns = int(seconds * 23400 * sps)
host = [workloads.bcch_carrier(pkg, 700 + a, seconds=seconds, sps=sps, stn=(5 * a) % 24, delay=a % 8, cfo_hz=40.0 * (a - 3),
                               esn0_db=10.0 + a)[0] for a in range(distinct)]
base = [torch.from_numpy(h).to(dev) for h in host]
slices = [base[a % distinct] for a in range(A)]
mine = pkg.shard.scatter_iq(slices, A, ns, src=0, device=dev)
rec, key = pkg.shard.rx_run_on_slices(pkg.api, mine, ns, sps=sps, device=dev, with_key=True)
out = pkg.shard.gather_records(rec, dst=0, device=dev, order_key=key)
print(bench._sharded_native(pkg, dist, dev, 0, 1, A, ns, distinct, sps, base, out))
dist.destroy_process_group()
