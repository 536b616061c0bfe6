"""Several host threads on one device (reference deployment: one gmr1_rx process per carrier file in parallel,
utils/gmr1_process_recording.py:103-110).  The calls that share the library's per-device workspace take turns -- the host
part under a per-device lock, the device part ordered by an event across streams (csrc/capi_common.h: WsLease) -- so
concurrent callers get exactly what they get alone."""
import threading

import numpy as np
import pytest

import workloads

pytestmark = pytest.mark.gpu


def _key(rec):
    return [(int(r["arfcn"]), int(r["chain"]), int(r["type"]), int(r["fn"]), int(r["tn"]), bytes(r["l2"])) for r in rec]


@pytest.mark.timeout(600)
def test_two_threads_share_the_device_workspace(gpu_api, orc, pkg):
    import torch
    # thread A: receive loop over two carriers on its own stream; thread B: FCCH sweeps + a long detection list + the
    # two-launch TCH3 call on another stream -- all workspace users, sizes chosen so that the workspace GROWS mid-way
    carriers = [workloads.bcch_carrier(pkg, 300 + a, seconds=2.5, sps=4, stn=3 * a, delay=a, cfo_hz=50.0 * a)[0] for a in range(2)]
    ns = carriers[0].size
    d_car = torch.from_numpy(np.concatenate(carriers).view(np.float32)).cuda()
    want_rx = [_key(orc.rx_run(c, sps=4, arfcn=a)[1]) for a, c in enumerate(carriers)]
    fc = workloads.fcch_streams(pkg, 48, seed=9)
    want_toa = np.array([orc.fcch_rough(fc["iq"][i], 4)[1] for i in range(6)])
    d_fc = torch.from_numpy(fc["iq"].view(np.float32)).cuda()
    d_fo = torch.from_numpy(fc["offset"].astype(np.int64)).cuda()
    nt = workloads.nt3_mix(pkg, 400, seed=3)
    sp = nt["speech"][:300]
    alone = gpu_api.tch3_rx_batch(nt["iq"], nt["offset"][sp], 474, sps=4, freq_shift=nt["freq_shift"][sp], want_ebits=False)
    errs, out = [], {}
    go = threading.Barrier(2)

    def thread_a():
        try:
            st = torch.cuda.Stream()
            go.wait()
            for it in range(6):
                rec, status, chains, found = gpu_api.rx_run_dev(st.cuda_stream, d_car.data_ptr(), [0, ns], [ns, ns], sps=4)
                assert not status.any()
                for a in range(2):
                    assert _key(rec[rec["arfcn"] == a]) == [(a,) + k[1:] for k in want_rx[a]], f"iteration {it}, carrier {a}"
            out["a"] = True
        except BaseException as e:      # noqa: BLE001 - reported by the main thread
            errs.append(("a", repr(e)))

    def thread_b():
        try:
            st = torch.cuda.Stream()
            toa = torch.zeros(48, dtype=torch.int32, device="cuda")
            rv = torch.zeros(48, dtype=torch.int32, device="cuda")
            go.wait()
            for it in range(6):
                n = 8 * (it + 1)                                     # growing batches: the workspace is re-allocated
                gpu_api.fcch_rough_batch_dev(st.cuda_stream, "fcch", n, 4, fc["n_samples"], d_fc.data_ptr(), d_fo.data_ptr(), None,
                                             toa.data_ptr(), rv.data_ptr())
                st.synchronize()
                assert np.array_equal(toa.cpu().numpy()[:6], want_toa), f"iteration {it}"
                got = gpu_api.tch3_rx_batch(nt["iq"], nt["offset"][sp], 474, sps=4, freq_shift=nt["freq_shift"][sp], want_ebits=False)
                for k in ("frame0", "frame1", "rv", "toa"):
                    assert np.array_equal(got[k], alone[k]), (it, k)
            out["b"] = True
        except BaseException as e:      # noqa: BLE001
            errs.append(("b", repr(e)))

    ts = [threading.Thread(target=thread_a), threading.Thread(target=thread_b)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(500)
    assert not errs, errs
    assert out.get("a") and out.get("b")
