"""Sharding of the receive path over the GPUs of one node (one process per GPU).

The units of this path are independent: pre-cut bursts (configs 3 / 5) never
interact except inside a FACCH3 group of 4, and ARFCNs (config 4) never interact at all.  So
the compute is sharded with NO data-path collective; the only exchanges are the two the
north star names, both outside the kernels:

  * scatter of per-ARFCN IQ slices from the rank that holds the capture (point-to-point
    send/recv, one peer per xGMI link -- not a ring),
  * gather of the fixed-size decoded-frame records back to that rank.

torch.distributed is plumbing here: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the
CPU tests (tests/test_shard.py runs world_size 2).
"""
from __future__ import annotations

import numpy as np

# decoded frame record, 40 bytes
RECORD_DTYPE = np.dtype([
    ("arfcn", "<u2"), ("chain", "u1"), ("type", "u1"), ("fn", "<u4"),
    ("tn", "u1"), ("crc", "u1"), ("len", "u1"), ("pad", "u1"),
    ("conv", "<i4"), ("l2", "u1", (24,)),
])
assert RECORD_DTYPE.itemsize == 40


def partition_contiguous(n_units: int, world: int, rank: int, group: int = 1):
    """Contiguous block [start, stop) of `n_units` for `rank`; block edges fall on multiples of
    `group` (FACCH3: 4 bursts form one frame and must stay on one GPU)."""
    n_groups = -(-n_units // group)
    per = -(-n_groups // world)
    g0 = min(rank * per, n_groups)
    g1 = min(g0 + per, n_groups)
    return min(g0 * group, n_units), min(g1 * group, n_units)


def owner_of_arfcn(arfcn_index: int, world: int) -> int:
    """ARFCN a -> rank a mod world (8 ARFCN per GPU for the 64-ARFCN capture of config 4)."""
    return arfcn_index % world


def my_arfcns(n_arfcn: int, world: int, rank: int):
    return [a for a in range(n_arfcn) if owner_of_arfcn(a, world) == rank]


def scatter_iq(slices, n_arfcn: int, n_samples: int, src: int = 0, device=None):
    """Rank `src` holds `slices[a]` (complex64 tensors of n_samples); every rank returns
    {arfcn: tensor} for the ARFCNs it owns.  Point-to-point isend/irecv, all in flight at once."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    mine = my_arfcns(n_arfcn, world, rank)
    out = {}
    ops = []
    if rank == src:
        for a in range(n_arfcn):
            r = owner_of_arfcn(a, world)
            t = torch.view_as_real(slices[a]).contiguous()
            if device is not None:
                t = t.to(device)
            if r == src:
                out[a] = torch.view_as_complex(t)
            else:
                ops.append(dist.P2POp(dist.isend, t, r, tag=a))
    else:
        bufs = {}
        for a in mine:
            bufs[a] = torch.empty((n_samples, 2), dtype=torch.float32, device=device)
            ops.append(dist.P2POp(dist.irecv, bufs[a], src, tag=a))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank != src:
        out = {a: torch.view_as_complex(b) for a, b in bufs.items()}
    return out


def gather_records(records: np.ndarray, dst: int = 0, device=None, order_key=None):
    """Gather variable-length record arrays (RECORD_DTYPE) on rank `dst`.

    Counts travel by all_gather (world x 8 bytes), payloads as padded fixed-size blocks (40-byte records plus an
    8-byte ordering key each).  The result on `dst` has the order a single gmr1_hip_rx_run over ALL carriers
    produces: carrier, then chain, then the order the chain emitted its frames in (frame numbers jump when a chain
    decodes its first SI1, gmr1_rx.c:194-233, so sorting by fn would be wrong).  `order_key[i]` =
    (global carrier index << 32) | position of record i in this rank's output; when it is not given the
    records' `arfcn` field is taken as the global carrier index (true when carriers are named 0..A-1).
    None on ranks other than `dst`."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    records = np.ascontiguousarray(records, dtype=RECORD_DTYPE)
    if order_key is None:
        order_key = (records["arfcn"].astype(np.uint64) << np.uint64(32)) | np.arange(records.size, dtype=np.uint64)
    order_key = np.ascontiguousarray(order_key, dtype=np.uint64)
    assert order_key.size == records.size
    cnt = torch.tensor([records.size], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    row = RECORD_DTYPE.itemsize + 8
    buf = np.zeros((cap, row), dtype=np.uint8)
    buf[:records.size, :RECORD_DTYPE.itemsize] = records.view(np.uint8).reshape(records.size, RECORD_DTYPE.itemsize)
    buf[:records.size, RECORD_DTYPE.itemsize:] = order_key.view(np.uint8).reshape(records.size, 8)
    payload = torch.from_numpy(buf)
    if device is not None:
        payload = payload.to(device)
    blocks = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(blocks, payload)      # < 0.5 MB in total for an ARFCN-minute: latency bound
    if rank != dst:
        return None
    rows = np.concatenate([blocks[r].cpu().numpy()[:counts[r]] for r in range(world)])
    allr = np.ascontiguousarray(rows[:, :RECORD_DTYPE.itemsize]).reshape(-1).view(RECORD_DTYPE)
    keys = np.ascontiguousarray(rows[:, RECORD_DTYPE.itemsize:]).reshape(-1).view(np.uint64)
    return allr[np.argsort(keys, kind="stable")]


def rx_run_on_slices(api, mine, n_samples: int, sps: int = 4, device=None, max_records: int = 1 << 16, arfcn_ids=None,
                     with_key: bool = False):
    """The receive loop (gmr1_hip_rx_run_dev, reference src/gmr1_rx.c:605-895) over the carriers this rank owns:
    `mine` = {global carrier index: complex64 tensor of n_samples}.  Returns the records (and, with `with_key`, the
    ordering key gather_records wants)."""
    import torch

    ids = sorted(mine)
    if not ids:
        rec = np.zeros(0, RECORD_DTYPE)
        return (rec, np.zeros(0, np.uint64)) if with_key else rec
    parts = [torch.view_as_real(mine[a]).reshape(-1) for a in ids]
    if device is not None:
        parts = [p.to(device) for p in parts]
    iq = torch.cat(parts).contiguous()
    offset = np.arange(len(ids), dtype=np.uint64) * np.uint64(n_samples)
    length = np.full(len(ids), n_samples, np.uint64)
    stream = torch.cuda.current_stream(iq.device).cuda_stream
    # the loop labels records with what it is given: the global carrier index, renamed afterwards if asked
    names = np.asarray(ids, np.uint16)
    rec, status, chains, found = api.rx_run_dev(stream, iq.data_ptr(), offset, length, sps=sps,
                                                arfcn=names, max_records=max_records)
    rec = np.array(rec, dtype=RECORD_DTYPE)
    key = (rec["arfcn"].astype(np.uint64) << np.uint64(32)) | np.arange(rec.size, dtype=np.uint64)
    if arfcn_ids is not None:
        rec["arfcn"] = np.asarray(arfcn_ids, np.uint16)[rec["arfcn"]]
    return (rec, key) if with_key else rec


def rx_capture_sharded(api, slices, n_arfcn: int, n_samples: int, sps: int = 4, src: int = 0, device=None,
                       max_records: int = 1 << 16, arfcn_ids=None):
    """BASELINE.md config 4 end to end on the ranks of one node: rank `src` holds the channelised
    capture (`slices[a]`, complex64 tensors of n_samples), every rank receives the ARFCNs it owns,
    runs the receive loop on them (gmr1_hip_rx_run_dev, reference src/gmr1_rx.c:605-895) and the
    decoded-frame records come back to `src`, in the order one gmr1_hip_rx_run over all carriers gives.
    No collective touches the data path."""
    mine = scatter_iq(slices, n_arfcn, n_samples, src=src, device=device)
    rec, key = rx_run_on_slices(api, mine, n_samples, sps=sps, device=device, max_records=max_records,
                                arfcn_ids=arfcn_ids, with_key=True)
    return gather_records(rec, dst=src, device=device, order_key=key)


def rx_wideband_sharded(api, wide, n_in: int, samp_rate: float, channels, sps: int = 4, src: int = 0,
                        device=None, max_records: int = 1 << 16):
    """BASELINE.json configs[3] from the wideband container: rank `src` holds the capture (`wide`, a
    device tensor of n_in complex64 samples viewed as float32 pairs; other ranks pass None), channelizes
    the requested raster positions (gmr1_hip_channelize_dev, reference utils/gmr1_rx_sdr.py:391-602), the
    per-ARFCN streams go to their owners point to point, every rank runs the receive loop on its share
    and the decoded frames come back to `src` (records carry the raster position as arfcn)."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank()
    channels = [int(c) for c in channels]
    _, _, n_out = api.channelize_plan(samp_rate, sps, n_in)
    slices = None
    if rank == src:
        out = torch.empty((len(channels), n_out, 2), dtype=torch.float32, device=wide.device)
        stream = torch.cuda.current_stream(wide.device).cuda_stream
        api.channelize_dev(stream, wide.data_ptr(), n_in, samp_rate, channels, out.data_ptr(), n_out, sps=sps)
        slices = [torch.view_as_complex(out[i]) for i in range(len(channels))]
    return rx_capture_sharded(api, slices, len(channels), n_out, sps=sps, src=src, device=device,
                              max_records=max_records, arfcn_ids=channels)
