"""GPU tests of the wideband -> per-ARFCN channelizer (gmr1_hip_channelize*, reference
utils/gmr1_rx_sdr.py:391-602) against the numpy restatement in oracle/orc_chan.py, and end to end:
a synthetic wideband capture of BCCH carriers -> channelizer -> receive loop -> the frames that were sent."""
import os
import sys

import numpy as np
import pytest

import workloads

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))

pytestmark = pytest.mark.gpu

FS = 2.0e6


def test_plan_matches_the_reference_formulas(gpu_api):
    import orc_chan
    pl = orc_chan.Plan(FS)
    n_chans, n_mid, n_out = gpu_api.channelize_plan(FS, 4, 200000)
    assert n_chans == pl.n_chans == 64 and n_mid == 200000 // 32
    assert n_out == ((n_mid * 32 - (pl.taps_resamp.size // 2) % 32) * 117) // 2500
    # not n_chans x 31.25 kHz: the pre-resampler comes first (2.048 Msps -> 66 channels, rate 4125 / 4096)
    po = orc_chan.Plan(2.048e6)
    n_chans, n_mid, n_out = gpu_api.channelize_plan(2.048e6, 4, 204800)
    n_pre = ((204800 * 32 - (po.taps_pre.size // 2) % 32) * 4125) // (32 * 4096)
    assert n_chans == po.n_chans == 66 and n_mid == n_pre // 33
    with pytest.raises(Exception):
        gpu_api.channelize_plan(1.9e6 + 0.5, 4, 1000)      # off the grid and not a whole number of Hz


def test_channelizer_matches_oracle(gpu_api):
    import orc_chan
    pl = orc_chan.Plan(FS)
    rng = np.random.default_rng(2)
    n = 300000
    x = (rng.standard_normal((n, 2)) * 0.3).astype(np.float32).view(np.complex64).reshape(-1)
    s = np.arange(n)
    for k, f, a in ((5, 1000.0, 1.0), (40, -4000.0, 0.5), (31, 9000.0, 2.0)):
        kk = k if k < 32 else k - 64
        x += (a * np.exp(2j * np.pi * ((kk * 31250.0 + f) / FS) * s)).astype(np.complex64)
    chans = [5, 40, 31, 0, 63, 32]
    got = gpu_api.channelize(x, FS, chans)
    ref = orc_chan.channelize(x, pl, chans)
    for i, k in enumerate(chans):
        r = ref[k]
        assert got[i].size == r.size
        err = np.max(np.abs(got[i] - r))
        assert err < 2e-4 * max(1.0, float(np.sqrt(np.mean(np.abs(r) ** 2)))), (k, err)
    # the tones came out where they belong, at the right frequency (analytic check of oracle and kernel alike)
    for i, (k, f, a) in enumerate(((5, 1000.0, 1.0), (40, -4000.0, 0.5), (31, 9000.0, 2.0))):
        z = got[i][2000:20000].astype(np.complex128)
        fest = np.angle(np.mean(z[1:] * np.conj(z[:-1]))) / (2 * np.pi) * 93600.0
        assert abs(fest - f) < 30.0, (k, fest, f)
        assert abs(np.sqrt(np.mean(np.abs(z) ** 2)) - a) < 0.25 * a
    # pre-rotation moves everything by one raster step
    rot = gpu_api.channelize(x, FS, [6], rotation=2 * np.pi * 31250.0 / FS)
    assert np.max(np.abs(rot[0][3000:] - got[0][3000:])) < 5e-3


@pytest.mark.parametrize("fs", [1.0e6, 1.25e6, 2.5e6, 4.0e6])
def test_other_channel_counts_match_oracle(gpu_api, fs):
    """Sample rates whose channel count is not 64 (gmr1_rx_sdr.py:408: 32, 40, 80, 128 channels) run the
    generic filterbank kernel; same oracle, same tolerance."""
    import orc_chan
    pl = orc_chan.Plan(fs)
    M = pl.n_chans
    assert M == int(round(fs / 31250.0)) and M != 64
    rng = np.random.default_rng(int(fs) % 1000 + 5)
    n = 40 * M * 50 + 37                                   # not a whole number of instants
    x = (rng.standard_normal((n, 2)) * 0.3).astype(np.float32).view(np.complex64).reshape(-1)
    s = np.arange(n)
    tones = ((3, 1500.0, 1.0), (M - 2, -2500.0, 0.7), (M // 2 - 1, 6000.0, 1.5))
    for k, f, a in tones:
        kk = k if k < M // 2 else k - M
        x += (a * np.exp(2j * np.pi * ((kk * 31250.0 + f) / fs) * s)).astype(np.complex64)
    chans = [3, M - 2, M // 2 - 1, 0, M - 1, M // 2]
    n_chans, n_mid, n_out = gpu_api.channelize_plan(fs, 4, n)
    assert n_chans == M and n_mid == n // (M // 2)
    got = gpu_api.channelize(x, fs, chans)
    ref = orc_chan.channelize(x, pl, chans)
    for i, k in enumerate(chans):
        r = ref[k]
        assert got[i].size == r.size == n_out
        err = np.max(np.abs(got[i] - r))
        assert err < 2e-4 * max(1.0, float(np.sqrt(np.mean(np.abs(r) ** 2)))), (fs, k, err)
    for i, (k, f, a) in enumerate(tones):
        z = got[i][1500:].astype(np.complex128)
        fest = np.angle(np.mean(z[1:] * np.conj(z[:-1]))) / (2 * np.pi) * 93600.0
        assert abs(fest - f) < 40.0, (fs, k, fest, f)
    # pre-rotation by one raster step: channel 4 of the rotated capture = channel 3 of the plain one
    rot = gpu_api.channelize(x, fs, [4], rotation=2 * np.pi * 31250.0 / fs)
    assert np.max(np.abs(rot[0][3000:] - got[0][3000:])) < 5e-3


def test_channelizer_device_resident_and_subset(gpu_api):
    import torch
    rng = np.random.default_rng(3)
    n = 64 * 1000 + 17                                      # ragged tail
    x = rng.standard_normal((n, 2)).astype(np.float32).view(np.complex64).reshape(-1)
    full = gpu_api.channelize(x, FS, list(range(64)))
    _, _, n_out = gpu_api.channelize_plan(FS, 4, n)
    t = torch.from_numpy(x.view(np.float32)).cuda()
    out = torch.zeros((2, n_out + 5, 2), dtype=torch.float32, device="cuda")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        w = gpu_api.channelize_dev(st.cuda_stream, t.data_ptr(), n, FS, [9, 50], out.data_ptr(), n_out + 5)
    st.synchronize()
    assert w == n_out
    o = out.cpu().numpy().view(np.complex64).reshape(2, n_out + 5)
    assert np.array_equal(o[0, :n_out], full[9]) and np.array_equal(o[1, :n_out], full[50])
    assert not o[:, n_out:].any()
    with pytest.raises(Exception):
        gpu_api.channelize(x, FS, [3, 3])


def test_wideband_capture_decodes_end_to_end(gpu_api, pkg):
    carriers = ((3, dict(stn=3, delay=2, cfo_hz=80.0)), (17, dict(stn=10, delay=5, cfo_hz=-150.0)),
                (60, dict(stn=0, delay=0, cfo_hz=20.0)))
    wide, sents = workloads.wideband_capture(pkg, 11, seconds=2.5, carriers=carriers)
    chans = [c for c, _ in carriers] + [30]                 # one raster position nobody transmits on
    nb = gpu_api.channelize(wide, FS, chans)
    n_out = nb.shape[1]
    assert abs(n_out - 2.5 * 93600) < 200
    offset = np.arange(len(chans), dtype=np.uint64) * np.uint64(n_out)
    length = np.full(len(chans), n_out, np.uint64)
    rec, status, chains, found = gpu_api.rx_run(nb.reshape(-1), offset, length, sps=4, arfcn=np.asarray(chans, np.uint16))
    for ch, _ in carriers:
        mine = rec[rec["arfcn"] == ch]
        mb, nbc, mc, nc, mp = workloads.match_records(mine, sents[ch])
        n_b = sum(s["type"] == "bcch" for s in sents[ch])
        n_c = sum(s["type"] == "ccch" for s in sents[ch])
        assert nbc >= n_b - 3 and mp == nbc, (ch, nbc, n_b, mp)
        assert nc >= 0.8 * n_c and mc >= nc - 1, (ch, nc, n_c, mc)
    assert len(rec[rec["arfcn"] == 30]) == 0


def test_wideband_sharded_single_rank(gpu_api, pkg):
    """configs[3] on a world of one: channelize -> scatter (local) -> receive loop -> gather."""
    import torch
    import torch.distributed as dist
    from importlib import import_module
    shard = import_module(pkg.__name__ + ".shard")
    carriers = ((5, dict(stn=2, delay=1, cfo_hz=50.0)), (58, dict(stn=12, delay=4, cfo_hz=-60.0)))
    wide, sents = workloads.wideband_capture(pkg, 21, seconds=2.0, carriers=carriers)
    w = torch.from_numpy(wide.view(np.float32)).cuda()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29543", rank=0, world_size=1)
    try:
        rec = shard.rx_wideband_sharded(gpu_api, w, wide.size, FS, [5, 58, 20])
    finally:
        dist.destroy_process_group()
    assert set(np.unique(rec["arfcn"])) == {5, 58}
    for ch, _ in carriers:
        mine = rec[rec["arfcn"] == ch]
        mb, nbc, mc, nc, mp = workloads.match_records(mine, sents[ch])
        assert nbc >= 3 and mp == nbc and mc >= nc - 1


# ---- direct mode (gmr1_rx_sdr.py:605-807): gmr1_hip_ddc* -----------------------------------------------------------
@pytest.mark.parametrize("fs", [2.0e6, 1.25e6, 2.5e6, 4.0e6, 1.0e6])
def test_direct_mode_matches_oracle(gpu_api, fs):
    """Frequency-translating FIR, second FIR, arbitrary resampler against the numpy restatement, at the decimation
    splits the script picks for these rates (7 x 6, 5 x 5, 7 x 7, 12 x 7; 1.0 Msps: 5 x 1 and a resampler that goes DOWN,
    rate 0.468, 95 taps per phase -- the kernel's long instantiation), carriers on and off the 31.25 kHz raster."""
    import orc_chan
    pl = orc_chan.DirectPlan(fs)
    d1, d2, rs, n_out = gpu_api.ddc_plan(fs, 4, 200000)
    assert (d1, d2) == (pl.decim1, pl.decim2) and abs(rs - pl.resamp) < 1e-12
    rng = np.random.default_rng(int(fs) % 97)
    n = 200000
    x = (rng.standard_normal((n, 2)) * 0.3).astype(np.float32).view(np.complex64).reshape(-1)
    s = np.arange(n)
    freqs = [3 * 31250.0, -7 * 31250.0 + 400.0, 0.0]
    for f, a in zip(freqs, (1.0, 0.6, 1.5)):
        x += (a * np.exp(2j * np.pi * ((f + 900.0) / fs) * s)).astype(np.complex64)
    got = gpu_api.ddc(x, fs, freqs)
    assert got.shape == (3, n_out)
    for i, f in enumerate(freqs):
        ref = orc_chan.direct_ddc(x, pl, f, n_out=n_out)
        err = np.max(np.abs(got[i] - ref))
        assert err < 2e-4 * max(1.0, float(np.sqrt(np.mean(np.abs(ref) ** 2)))), (fs, f, err)
        z = got[i][1000:9000].astype(np.complex128)
        fest = np.angle(np.mean(z[1:] * np.conj(z[:-1]))) / (2 * np.pi) * 93600.0
        assert abs(fest - 900.0) < 40.0, (fs, f, fest)


def test_direct_mode_refusals(gpu_api):
    with pytest.raises(gpu_api.Gmr1HipError, match="-22"):
        gpu_api.ddc_plan(93600.0 * 20, 4, 1000)           # the reference's own exact case cannot run
    # (no plan's resampler is refused for its length any more: the longest, rate 0.468 at 0.8 / 1.0 Msps, has 95 taps per
    # phase and the kernel's long instantiation holds 96)
    d1, d2, rs, _ = gpu_api.ddc_plan(1.5e6, 4, 100000)
    assert (d1, d2) == (8, 1) and abs(rs - 0.4992) < 1e-9


def test_direct_mode_decodes_end_to_end(gpu_api, pkg):
    """Wideband capture -> direct branches of its three carriers -> receive loop: the frames that were sent (the same
    capture the filterbank test decodes)."""
    wide, sents = workloads.wideband_capture(pkg, 21, seconds=2.5)
    chans = sorted(sents)
    freqs = [(c if c < 32 else c - 64) * 31250.0 for c in chans]
    streams = gpu_api.ddc(wide, FS, freqs)
    n = streams.shape[1]
    iq = streams.reshape(-1)
    rec, status, chains, found = gpu_api.rx_run(iq, np.arange(len(chans), dtype=np.uint64) * n, np.full(len(chans), n, np.uint64),
                                                sps=4, arfcn=np.array(chans, np.uint16))
    assert list(status) == [0] * len(chans)
    for c in chans:
        mine = rec[rec["arfcn"] == c]
        mb, nbc, mc, nc, mp = workloads.match_records(mine, sents[c])
        n_b = sum(s["type"] == "bcch" for s in sents[c])
        n_c = sum(s["type"] == "ccch" for s in sents[c])
        assert nbc >= n_b - 3 and mp == nbc, (c, nbc, n_b, mp)
        assert nc >= 0.7 * n_c and mc >= nc - 1, (c, nc, n_c, mc)     # (the three filters' start-up costs the first frames)


@pytest.mark.parametrize("fs", [2.048e6, 1.92e6, 2.4e6])
def test_off_grid_sample_rates_pre_resampler(gpu_api, fs):
    """Sample rates off the 31.25 kHz grid (gmr1_rx_sdr.py:413-417, :453-461): the capture is resampled to n_chans x 31.25
    kHz by the 32-phase pre-resampler, then channelized.  Against the numpy oracle, and analytically: a tone placed on
    ARFCN k (+ an offset) at the off-grid rate comes out of channel k at that offset at 4 samples per symbol; with a
    pre-rotation of one raster step it comes out of channel k + 1."""
    import orc_chan
    pl = orc_chan.Plan(fs)
    assert pl.pre_rate is not None
    M = pl.n_chans
    rng = np.random.default_rng(int(fs) % 1000)
    n = int(0.12 * fs)
    x = (rng.standard_normal((n, 2)) * 0.1).astype(np.float32).view(np.complex64).reshape(-1)
    s = np.arange(n)
    # (carriers beyond 0.4 x the sample rate from the centre sit in the pre-resampler's transition band, as with GNU
    # Radio's default prototype: none of the test tones does)
    tones = ((5, 1000.0, 1.0), (M - 9, -4000.0, 0.5), (M // 4 + 3, 6000.0, 1.5))
    for k, f, a in tones:
        kk = k if k < M // 2 else k - M
        x += (a * np.exp(2j * np.pi * ((kk * 31250.0 + f) / fs) * s)).astype(np.complex64)
    chans = [t[0] for t in tones] + [0, M - 1]
    got = gpu_api.channelize(x, fs, chans)
    ref = orc_chan.channelize(x, pl, chans)
    for i, k in enumerate(chans):
        r = ref[k]
        assert got[i].size == r.size and abs(r.size - n / fs * 93600) < 40
        err = np.max(np.abs(got[i] - r))
        assert err < 2e-4 * max(1.0, float(np.sqrt(np.mean(np.abs(r) ** 2)))), (k, err)
    for i, (k, f, a) in enumerate(tones):
        z = got[i][2000:9000].astype(np.complex128)
        fest = np.angle(np.mean(z[1:] * np.conj(z[:-1]))) / (2 * np.pi) * 93600.0
        assert abs(fest - f) < 30.0, (k, fest, f)
        assert abs(np.sqrt(np.mean(np.abs(z) ** 2)) - a) < 0.25 * a
    rot = float(np.float32(2 * np.pi * 31250.0 / fs))          # the C API takes the rotation as a float
    shifted = gpu_api.channelize(x, fs, [6], rotation=rot)
    ref_rot = orc_chan.channelize(x, pl, [6], rotation=rot)[6]
    assert np.max(np.abs(shifted[0] - ref_rot)) < 2e-4 * max(1.0, float(np.sqrt(np.mean(np.abs(ref_rot) ** 2))))
    # one raster step up: channel 6 now carries what channel 5 carried -- up to the constant phase the rotation picks up
    # over the pre-resampler's delay (the rotator runs at the capture's rate, the filterbank's mixers after the resampler)
    a, b = got[0][3000:].astype(np.complex128), shifted[0][3000:].astype(np.complex128)
    c = np.vdot(a, b)
    c /= abs(c)
    assert np.max(np.abs(b - c * a)) < 5e-3


def test_channelizer_planar_output_feeds_planar_receive(gpu_api, pkg):
    """gmr1_hip_channelize_planar_dev writes the streams polyphase-planar; the same samples as the interleaved call, at
    out_planes[(g % sps) * plane_stride + g // sps] for flat index g = stream * out_stride + m."""
    import torch
    rng = np.random.default_rng(8)
    n = 200000
    x = rng.standard_normal((n, 2)).astype(np.float32)
    t = torch.from_numpy(x).cuda()
    chans = [3, 60, 17]
    for sps in (4, 3):
        _, _, n_out = gpu_api.channelize_plan(FS, sps, n)
        stride = n_out + 7
        flat = torch.zeros((len(chans), stride, 2), dtype=torch.float32, device="cuda")
        gpu_api.channelize_dev(None, t.data_ptr(), n, FS, chans, flat.data_ptr(), stride, sps=sps)
        P = -(-len(chans) * stride // sps) + 2
        planes = torch.zeros((sps * P, 2), dtype=torch.float32, device="cuda")
        w = gpu_api.channelize_planar_dev(None, t.data_ptr(), n, FS, chans, planes.data_ptr(), stride, P, sps=sps)
        assert w == n_out
        torch.cuda.synchronize()
        a = flat.cpu().numpy().view(np.complex64).reshape(len(chans), stride)
        b = planes.cpu().numpy().view(np.complex64).reshape(-1)
        g = (np.arange(len(chans))[:, None] * stride + np.arange(n_out)[None, :])
        assert np.array_equal(b[(g % sps) * P + g // sps], a[:, :n_out])
        assert np.abs(a[:, :n_out]).max() > 0
    with pytest.raises(Exception):
        gpu_api.channelize_planar_dev(None, t.data_ptr(), n, FS, chans, planes.data_ptr(), stride, 10, sps=4)


def test_non_finite_sample_poisons_only_the_instants_its_taps_reach(gpu_api):
    """A clipped / corrupt capture: one sample with an Inf in one component and a NaN in the other.  The filterbank's
    select-free DFT stages spread a non-finite component over both components of what they touch (chan_kernels.hip:
    pf_stage) -- but only over the output instants whose filter taps reach that sample: everything before and well after
    it is bit-identical to the clean capture's output."""
    rng = np.random.default_rng(8)
    n = 400000
    x = (rng.standard_normal((n, 2)) * 0.3).astype(np.float32).view(np.complex64).reshape(-1)
    bad = x.copy()
    k = n // 2
    bad[k] = np.complex64(complex(np.inf, np.nan))
    chans = [3, 40, 63]
    good = gpu_api.channelize(x, FS, chans)
    got = gpu_api.channelize(bad, FS, chans)
    for g, b in zip(good, got):
        m = g.size
        centre = int(k * m / n)
        # filter spans: 1025-tap prototype at 2 Msps + the resampler's 11 symbols -- a few hundred output samples at most
        lo, hi = centre - 400, centre + 400
        assert np.array_equal(g[:lo], b[:lo]) and np.array_equal(g[hi:], b[hi:])
        assert not np.isfinite(b[lo:hi]).all()
        assert np.isfinite(b[:lo]).all() and np.isfinite(b[hi:]).all()


def test_resampler_with_one_ring_per_work_group_is_bit_identical(gpu_api):
    """k_resamp2w (the profiling build's GMR1_HIP_RESAMP_WG=1: the eight waves of a period share one window ring, fetched once,
    released to each other through counters in LDS) against k_resamp2 in the product build: the same streams bit for bit, stream
    ends included (windows that cross them, periods past the last output).  A child process on the profiling library."""
    import subprocess
    rng = np.random.default_rng(11)
    n = 420000
    x = (rng.standard_normal((n, 2)) * 0.5).astype(np.float32).view(np.complex64).reshape(-1)
    chans = [0, 7, 31, 32, 63]
    got = gpu_api.channelize(x, FS, chans)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "osmo-gmr_amd", "libgmr1_hip_prof.so")
    if not os.path.exists(prof):
        pytest.skip("the profiling build (python osmo-gmr_amd/build.py --profile) is not there")
    np.save("/tmp/_rsw_in.npy", x)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from __graft_entry__ import load_package; pkg = load_package(); "
            "import torch; torch.cuda.init(); pkg.api.load(); pkg.api.init(0); x = np.load('/tmp/_rsw_in.npy'); "
            "y = pkg.api.channelize(x, %r, %r); np.save('/tmp/_rsw_out.npy', np.stack(y))" % (root, FS, chans))
    env = dict(os.environ, GMR1_HIP_LIBRARY=prof, GMR1_HIP_RESAMP_WG="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    other = np.load("/tmp/_rsw_out.npy")
    assert other.shape == np.stack(got).shape
    assert np.array_equal(other.view(np.uint32), np.stack(got).view(np.uint32))
