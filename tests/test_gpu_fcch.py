"""GPU parity tests of the FCCH acquisition kernels vs the CPU oracle (config 2)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SPS = 4


def _streams(pkg, n, n_samples, seed, snr_db=(0.0, 6.0), cfo=2000.0):
    rng = np.random.default_rng(seed)
    out = np.zeros((n, n_samples), np.complex64)
    starts = []
    for i in range(n):
        x, st = pkg.synth.synth_fcch_stream(n_samples, SPS, rng, snr_db=float(rng.choice(snr_db)),
                                            cfo_hz=float(rng.uniform(-cfo, cfo)))
        out[i] = x
        starts.append(st)
    return out, starts


def test_fcch_rough_batch_matches_oracle(gpu_api, orc, pkg):
    """1-s streams (93 600 samples): full 23 284-lag sweep, toa identical to the oracle."""
    n, ns = 12, 93600
    x, starts = _streams(pkg, n, ns, seed=2)
    offset = (np.arange(n) * ns).astype(np.uint64)
    toa, rv = gpu_api.fcch_rough_batch(x, offset, ns, sps=SPS)
    assert not rv.any()
    for i in range(n):
        orv, otoa = orc.fcch_rough(x[i], SPS)
        assert orv == 0
        assert toa[i] == otoa, (i, toa[i], otoa)
        # and it is a real FCCH position: a CFO of f Hz moves the chirp correlation peak by
        # f / 2995.2 ms (fcch.c:611), i.e. up to 16 symbols at 2 kHz; the fine stage removes that
        assert min(abs(toa[i] - s) for s in starts[i]) <= 20 * SPS


def test_fcch_rough_window_lengths_and_freq_shift(gpu_api, orc, pkg):
    """gmr1_rx's 330 ms window (30 888 samples), ragged lengths, non-zero freq_shift."""
    rng = np.random.default_rng(5)
    for ns in (30888, 30001, 4 * 117 + 40, 50000):
        x, _ = pkg.synth.synth_fcch_stream(ns, SPS, rng, snr_db=6.0, cfo_hz=300.0, first=min(1000, ns // 4))
        for fs in (0.0, -0.05):
            toa, rv = gpu_api.fcch_rough_batch(x, np.zeros(1, np.uint64), ns, sps=SPS,
                                               freq_shift=np.array([fs], np.float32))
            orv, otoa = orc.fcch_rough(x, SPS, fs)
            assert rv[0] == 0 and orv == 0 and toa[0] == otoa, (ns, fs, toa[0], otoa)
    # the reference's own single call
    rv, toa = gpu_api.fcch_rough(x, SPS, 0.0)
    assert rv == 0 and toa == orc.fcch_rough(x, SPS)[1]


def test_fcch_fine_and_snr_match_oracle(gpu_api, orc, pkg):
    rng = np.random.default_rng(7)
    n = 40
    bursts = np.zeros((n, 468), np.complex64)
    for i in range(n):
        cfo = float(rng.uniform(-1500, 1500))
        x, _ = pkg.synth.synth_fcch_stream(468 + 64, SPS, rng, snr_db=float(rng.choice([0.0, 6.0, 20.0])),
                                           cfo_hz=cfo, first=int(rng.integers(0, 40)), period_sym=10000)
        bursts[i] = x[16:16 + 468]
    offset = (np.arange(n) * 468).astype(np.uint64)
    toa, fe = gpu_api.fcch_fine_batch(bursts, offset, sps=SPS)
    snr = gpu_api.fcch_snr_batch(bursts, offset, sps=SPS)
    for i in range(n):
        rv, otoa, ofe = orc.fcch_fine(bursts[i], SPS)
        assert rv == 0
        assert toa[i] == otoa, (i, toa[i], otoa)
        assert abs(fe[i] - ofe) < 1e-4, (i, fe[i], ofe)          # rad / symbol
        rv, osnr = orc.fcch_snr(bursts[i], SPS)
        assert rv == 0 and abs(snr[i] - osnr) <= 2e-4 * max(1.0, abs(osnr)), (i, snr[i], osnr)
    # single calls + the length check of the reference (-EINVAL)
    rv, t, f = gpu_api.fcch_fine(bursts[0], SPS)
    assert rv == 0 and t == toa[0] and abs(f - fe[0]) < 1e-6
    assert gpu_api.fcch_fine(bursts[0][:400], SPS)[0] == -22
    rv, s = gpu_api.fcch_snr(bursts[1], SPS)
    assert rv == 0 and abs(s - snr[1]) <= 1e-5 * max(1.0, abs(s))
    assert gpu_api.fcch_snr(bursts[0][:400], SPS)[0] == -22


def test_fcch3_long_chirp(gpu_api, orc, pkg):
    """FCCH3 (468-symbol chirp): rough + fine on one stream."""
    rng = np.random.default_rng(9)
    ns = 20000
    x, st = pkg.synth.synth_fcch_stream(ns, SPS, rng, snr_db=6.0, cfo_hz=100.0, first=3000, period_sym=100000,
                                        freq=0.32, length=468)
    toa, rv = gpu_api.fcch_rough_batch(x, np.zeros(1, np.uint64), ns, sps=SPS, fcch_type="fcch3_lband")
    orv, otoa = orc.fcch_rough(x, SPS, which="fcch3_lband")
    assert rv[0] == 0 and orv == 0 and toa[0] == otoa
    b = x[otoa:otoa + 468 * SPS]
    t, fe = gpu_api.fcch_fine_batch(b, np.zeros(1, np.uint64), sps=SPS, fcch_type="fcch3_lband")
    rv, ot, ofe = orc.fcch_fine(b, SPS, which="fcch3_lband")
    assert rv == 0 and t[0] == ot and abs(fe[0] - ofe) < 1e-4


def test_fcch_rough_multi(gpu_api, orc, pkg):
    """gmr1_fcch_rough_multi on 650 ms windows with 1-3 overlapping FCCH trains (fcch_multi_process,
    gmr1_rx.c:658-664): ranked peak list identical to the oracle."""
    rng = np.random.default_rng(13)
    ns = 60840
    n = 6
    x = np.zeros((n, ns), np.complex64)
    for i in range(n):
        s, _ = pkg.synth.synth_fcch_stream(ns, SPS, rng, snr_db=6.0, cfo_hz=float(rng.uniform(-300, 300)),
                                           first=int(rng.integers(200, 5000)))
        for extra in range(i % 3):                       # overlay weaker trains at other offsets
            s2, _ = pkg.synth.synth_fcch_stream(ns, SPS, rng, snr_db=3.0, cfo_hz=float(rng.uniform(-300, 300)),
                                                first=int(rng.integers(8000, 25000)))
            s = s + 0.7 * s2
        x[i] = s
    offset = (np.arange(n) * ns).astype(np.uint64)
    cnt, toa = gpu_api.fcch_rough_multi_batch(x, offset, ns, sps=SPS, N=16)
    for i in range(n):
        orv, otoa = orc.fcch_rough_multi(x[i], SPS, N=16)
        assert cnt[i] == orv, (i, cnt[i], orv)
        assert list(toa[i, :max(orv, 0)]) == list(otoa), (i, toa[i], otoa)
        assert orv >= 1
    # single call, too-short window (-EINVAL like the reference), pure noise (no 320 ms periodicity)
    rv, t = gpu_api.fcch_rough_multi(x[2], SPS, 0.0, 16)
    assert rv == cnt[2] and list(t) == list(toa[2, :rv])
    assert gpu_api.fcch_rough_multi(x[0][:30000], SPS)[0] == -22
    noise = (rng.standard_normal(ns) + 1j * rng.standard_normal(ns)).astype(np.complex64)
    g = gpu_api.fcch_rough_multi_batch(noise, np.zeros(1, np.uint64), ns, sps=SPS)[0][0]
    assert g == orc.fcch_rough_multi(noise, SPS)[0]


def test_folded_sweep_fallback_and_dc_offset(gpu_api, orc, pkg):
    """The folded rough sweep (k_fcch_sweep<NT, true>: a tile finishes its own lags once every tile of the stream has published
    its statistics) and its fallback.  (1) Streams with a DC offset much larger than the signal -- the mean's contribution to
    every lag, the one thing a tile cannot know by itself, decides the peak here -- give the oracle's toa.  (2) The profiling
    build with GMR1_HIP_FCCH_FOLD_POLLS=0 (every tile gives up at once and k_fcch_energy finishes the old way) and with
    GMR1_HIP_FCCH_UNFOLDED=1 (the two-kernel form) give the same toa, in a child process each."""
    import json
    import os
    import subprocess
    import sys
    n, ns = 44, 93600                                # 44 x 12 tiles: a launch large enough to be folded (> 512 work-groups)
    x, _ = _streams(pkg, n, ns, seed=23)
    x[1] += np.complex64(7.0 - 3.0j)                 # noise deviation ~ 1
    x[2] += np.complex64(-40.0 + 25.0j)
    x[4] *= np.float32(1e-3)
    x[43] += np.complex64(3.0 + 11.0j)
    offset = (np.arange(n) * ns).astype(np.uint64)
    toa, rv = gpu_api.fcch_rough_batch(x, offset, ns, sps=SPS)
    assert not rv.any()
    want = [orc.fcch_rough(x[i], SPS) for i in range(n)]
    assert all(w[0] == 0 for w in want)
    assert list(toa) == [w[1] for w in want]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "osmo-gmr_amd", "libgmr1_hip_prof.so")
    if not os.path.exists(prof):
        pytest.skip("the profiling build (python osmo-gmr_amd/build.py --profile) is not there")
    np.save("/tmp/_fold_streams.npy", x)
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r); from __graft_entry__ import load_package; pkg = load_package(); "
            "import torch; torch.cuda.init(); pkg.api.load(); pkg.api.init(0); x = np.load('/tmp/_fold_streams.npy'); "
            "toa, rv = pkg.api.fcch_rough_batch(x, (np.arange(%d) * %d).astype(np.uint64), %d, sps=%d); "
            "print(json.dumps([int(t) for t in toa] + [int(r) for r in rv]))" % (root, n, ns, ns, SPS))
    for switch in ("GMR1_HIP_FCCH_FOLD_POLLS", "GMR1_HIP_FCCH_UNFOLDED"):
        env = dict(os.environ, GMR1_HIP_LIBRARY=prof)
        env[switch] = "0" if switch.endswith("POLLS") else "1"
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        got = json.loads(r.stdout.strip().splitlines()[-1])
        assert got == [int(t) for t in toa] + [0] * n, (switch, got)
