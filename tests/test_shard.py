"""Multi-process CPU tests (gloo, world_size 2) of the sharding plumbing: partitioning,
IQ scatter from rank 0, record gather.  The per-rank "decode" is a stand-in (a checksum of
the received slice) because there is no GPU here; on the GPU box the same functions run
over RCCL."""
import os
import socket

import numpy as np
import pytest


def test_partitions(pkg):
    sh = pkg.shard
    # contiguous blocks cover everything exactly once, aligned to FACCH3 groups of 4
    for n, world, group in ((100000, 8, 1), (1000003, 8, 4), (10, 4, 4), (7, 8, 1), (0, 2, 4)):
        seen = np.zeros(n, int)
        for r in range(world):
            a, b = sh.partition_contiguous(n, world, r, group)
            assert 0 <= a <= b <= n
            if b < n:
                assert a % group == 0 and b % group == 0
            seen[a:b] += 1
        assert (seen == 1).all()
    assert sh.my_arfcns(64, 8, 3) == list(range(3, 64, 8))
    assert sorted(sum((sh.my_arfcns(64, 8, r) for r in range(8)), [])) == list(range(64))
    assert sh.RECORD_DTYPE.itemsize == 40


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _slice(a, n):
    rng = np.random.default_rng(1000 + a)
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)


def _worker(rank, world, port, n_arfcn, n_samples, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    sh = load_package().shard
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        slices = [torch.from_numpy(_slice(a, n_samples)) for a in range(n_arfcn)] if rank == 0 else None
        got = sh.scatter_iq(slices, n_arfcn, n_samples, src=0)
        assert sorted(got) == sh.my_arfcns(n_arfcn, world, rank)
        recs = np.zeros(0, sh.RECORD_DTYPE)
        for a, t in got.items():
            x = t.numpy()
            assert np.array_equal(x, _slice(a, n_samples)), f"ARFCN {a} arrived corrupted"
            # stand-in decode: (a % 3) + 1 records per ARFCN carrying a checksum of the slice
            k = (a % 3) + 1
            r = np.zeros(k, sh.RECORD_DTYPE)
            r["arfcn"] = a
            r["fn"] = np.arange(k)[::-1]
            r["tn"] = 5
            r["len"] = 24
            r["conv"] = int(np.abs(x).sum()) & 0x7fffffff
            r["l2"][:, 0] = a
            recs = np.concatenate([recs, r])
        out = sh.gather_records(recs, dst=0)
        if rank == 0:
            q.put(out)
        else:
            assert out is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_scatter_gather_world2(pkg):
    import torch.multiprocessing as mp
    world, n_arfcn, n_samples = 2, 5, 4096
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_arfcn, n_samples, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=150)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # every ARFCN's records arrived once, ordered by carrier and, inside a carrier, in the order the owning rank
    # emitted them (the stand-in emits descending frame numbers: a sort by fn would have reversed them) -- the order
    # one gmr1_hip_rx_run over all carriers produces
    exp_n = sum((a % 3) + 1 for a in range(n_arfcn))
    assert out.size == exp_n
    assert list(out["arfcn"]) == sorted(out["arfcn"])
    for a in range(n_arfcn):
        m = out[out["arfcn"] == a]
        assert m.size == (a % 3) + 1
        assert list(m["fn"]) == list(range((a % 3) + 1))[::-1]
        assert (m["l2"][:, 0] == a).all()
        assert (m["conv"] == (int(np.abs(_slice(a, n_samples)).sum()) & 0x7fffffff)).all()
