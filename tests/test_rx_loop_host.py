"""CPU test of the receive loop's control logic (osmo-gmr_amd/csrc/rx_loop.h, the code k_rx_chain runs on the
GPU): compiled for the host and walked through whole captures, against a Python model written from the
reference's process_bcch / burst_map / bcch_tdma_align (src/gmr1_rx.c:149-170, 194-233, 852-895)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(sps, length, align, fn, delay, stn, toa_step):
    """process_bcch frame by frame; a 'round' closes at a BCCH burst or after seven CCCH bursts"""
    frame_len = sps * 24 * 39
    out, rnd, frames, cnt, fb = [], 0, 0, 0, 0
    done = False                                   # c.done only ever becomes true in advance()

    def burst_map(tn, win):
        etoa = win >> 1
        b = align + sps * tn * 39 - etoa
        if b < 0 or b + 234 * sps + win > length:
            return None
        return b, etoa

    def advance():
        nonlocal fn, align, done, frames
        frames += 1
        fn += 1
        align += frame_len
        if align + 2 * frame_len > length:
            done = True

    items_in_round = 0
    while not done:
        sirfn = (fn - delay) & 63
        if sirfn % 8 == 2:
            m = burst_map(stn, 20 * sps)
            if m is not None:
                out.append((rnd, 1, m[0], fn, stn, m[1], fb))
                # rx_bcch: found + CRC ok -> feedback, then an SI1 every fourth round
                align += toa_step
                if (rnd & 3) == 1:
                    nd, ns = (rnd >> 2) & 7, (3 * rnd) % 24
                    sf, mf = rnd & 0x1fff, rnd & 3
                    nfn = (sf << 6) | (mf << 4) | (1 << 3) | ((2 + nd) & 7)
                    align += (stn - ns) * 39 * sps
                    fn, delay, stn = nfn, nd, ns
                advance()
                rnd, cnt, fb, items_in_round = rnd + 1, 0, 0, 0
                continue
        elif sirfn % 8 != 0:
            m = burst_map(stn, 10 * sps)
            if m is not None:
                out.append((rnd, 0, m[0], fn, stn, m[1], fb))
                cnt += 1
                items_in_round += 1
        advance()
        fb += 1
        if cnt == 7:
            rnd, cnt, fb, items_in_round = rnd + 1, 0, 0, 0
    if items_in_round:
        rnd += 1
    return out, (rnd, frames, align, fn, delay, stn)


@pytest.mark.parametrize("args", [
    (4, 4 * 23400 * 20, 9000, 0, 0, 0, 0),          # 20 s, nothing moves
    (4, 4 * 23400 * 12, 700, 5, 3, 7, 1),           # starts before the first slot fits; timing drifts
    (4, 4 * 23400 * 6, 30000, 61, 7, 23, -2),       # frame number wraps the 64-cycle, last slot
    (5, 5 * 23400 * 8, 1234, 2, 1, 11, 3),          # another oversampling
    (8, 8 * 23400 * 3, 0, 0, 0, 0, 0),
    (4, 4000, 100, 0, 0, 0, 0),                     # shorter than two frames: nothing to do
])
def test_round_listing_matches_the_frame_by_frame_loop(tmp_path, args):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "rx_loop_host")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror",
                           "-I" + os.path.join(ROOT, "osmo-gmr_amd", "csrc"),
                           os.path.join(ROOT, "tests", "c", "rx_loop_host.cpp"), "-o", exe])
    res = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, check=True)
    lines = res.stdout.strip().split("\n")
    got = [tuple(int(v) for v in ln.split()) for ln in lines[:-1]]
    end = tuple(int(v) for v in lines[-1].split()[1:])
    want, want_end = _model(*args)
    assert got == want
    assert end == want_end
