"""CPU tests of the channelizer oracle (oracle/orc_chan.py, reference utils/gmr1_rx_sdr.py:391-602).
GNU Radio is not available, so the restatement is pinned by construction: the numbers the script derives,
filter properties the firdes formulas guarantee, and a tone that has to come out of the right channel at
the right frequency."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))


def test_plan_numbers_of_the_reference_script():
    import orc_chan
    pl = orc_chan.Plan(2.0e6)
    assert pl.n_chans == 64 and pl.decim == 32                       # (ceil(2e6 / 31250) + 1) & ~1, :408
    assert pl.taps.size == 617                                       # int(53 fs / (22 tw)) | 1
    assert abs(pl.resamp - 1.4976) < 1e-12                           # 93.6 k / 62.5 k, :522
    assert pl.taps_resamp.size == 941                                # int(11 * 32 * 62500 / 23400) | 1, :523-529
    assert pl.freq2index(0.0) == 0 and pl.freq2index(3 * 31250.0) == 3 and pl.freq2index(-2 * 31250.0) == 62
    assert pl.freq2index(32 * 31250.0) is None and pl.freq2index(-32 * 31250.0) is None


def test_filter_design_properties():
    import orc_chan
    pl = orc_chan.Plan(2.0e6)
    h = pl.taps.astype(np.float64)
    assert np.allclose(h, h[::-1], atol=1e-9) and abs(h.sum() - 1.0) < 1e-5          # linear phase, unity DC gain
    H = np.abs(np.fft.rfft(h, 1 << 16))
    f = np.fft.rfftfreq(1 << 16, 1 / 2.0e6)
    assert H[np.searchsorted(f, 10000.0)] > 0.98 and H[np.searchsorted(f, 15625.0)] > 0.4
    assert H[np.searchsorted(f, 24000.0):].max() < 10 ** (-45 / 20)                  # Hamming: ~53 dB stop band
    r = pl.taps_resamp.astype(np.float64)
    assert np.allclose(r, r[::-1], atol=1e-7) and abs(r.sum() - 32.0) < 1e-3
    # root-raised cosine: matched with itself it is Nyquist at the symbol rate (32 * 62500 / 23400 samples per symbol)
    rc = np.convolve(r, r) / 32.0
    spb = 32.0 * 62500.0 / 23400.0
    c = rc.size // 2
    for m in (1, 2, 3, 4):
        v = np.interp(c + m * spb, np.arange(rc.size), rc)
        assert abs(v) < 0.02 * rc[c]


@pytest.mark.parametrize("fs", [2.0e6, 1.25e6, 2.5e6])
def test_tone_lands_in_its_channel_at_its_frequency(fs):
    import orc_chan
    pl = orc_chan.Plan(fs)
    M = pl.n_chans
    assert M == int(round(fs / 31250.0))               # 64, 40, 80 channels (gmr1_rx_sdr.py:408)
    n = int(0.06 * fs)
    s = np.arange(n)
    for k, f in ((7, 2500.0), (M - 4, -3000.0)):
        kk = k if k < M // 2 else k - M
        x = np.exp(2j * np.pi * ((kk * 31250.0 + f) / fs) * s).astype(np.complex64)
        y = orc_chan.pfb_channelizer_2x(x, pl.taps, pl.n_chans)
        p = np.mean(np.abs(y[:, 100:]) ** 2, axis=1)
        assert int(np.argmax(p)) == k and p[k] > 0.9
        others = np.delete(p, [k, (k + 1) % M, (k - 1) % M])
        assert others.max() < 1e-4
        z = orc_chan.arb_resampler(y[k], pl.resamp, pl.taps_resamp)
        assert abs(z.size - n / fs * 93600) < 8
        zz = z[600:4000].astype(np.complex128)
        fest = np.angle(np.mean(zz[1:] * np.conj(zz[:-1]))) / (2 * np.pi) * 93600.0
        assert abs(fest - f) < 5.0 and abs(np.mean(np.abs(zz)) - 1.0) < 0.05
