"""AMBE speech decoder on the GPU (ambe_kernels.hip through the C ABI) against the reference's outputs and the oracle.

Parity bar: **bit-identical samples**.  The kernel performs the reference's float operations in the
reference's order; what the reference gets from libm is either tabulated by the host's libm (enumerable arguments) or
computed the way glibc computes it (csrc/ambe_libm.h: powf, cosf), and tests/test_codec_host.py checks that restatement
against the host's libm bit for bit.  Every comparison below therefore demands equality; the count of differing
samples is printed with -s (before the libm restatement it was 1 in 2 * 10^5, each off by one step)."""
import os

import numpy as np
import pytest

import ambe_streams as S
import oracle_lib

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ambe_vectors.npz")


@pytest.fixture(scope="module")
def api(gpu_api):
    return gpu_api


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def compare(got, want, what):
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    n = int((d != 0).sum())
    print("%s: %d of %d samples differ, max |d| = %d" % (what, n, d.size, int(d.max()) if d.size else 0))
    assert n == 0, what
    return n


@pytest.mark.parametrize("name", ["random", "speech", "mixed", "unprimed"])
def test_matches_the_reference_program(api, gold, name):
    fr = gold[name + "_frames"]
    pcm, rv, _ = api.codec_decode_batch(fr[None])
    assert (rv[0] == gold[name + "_rv"]).all()
    if name == "unprimed":
        # the program's first frames show its start-up stack (decision D9): compare with the oracle, which starts from zeros
        want, _ = oracle_lib.ambe_decode(fr)
    else:
        want = gold[name + "_pcm"]
    compare(pcm[0], want, name)


@pytest.mark.parametrize("name", ["random", "speech", "mixed", "mixed_invalid"])
def test_cleared_matches_the_reference_on_a_clean_stack(api, gold, name):
    fr = gold[name + "_frames"]
    pcm, rv, _ = api.codec_decode_batch(fr[None], flags=api.CODEC_CLEARED)
    assert (rv[0] == gold[name + "_rv"]).all()
    compare(pcm[0], gold[name + "_clean"], name + " cleared")


def test_many_channels_against_the_oracle(api):
    n_ch, n_fr = 96, 150
    fr = np.stack([S.mixed_stream(n_fr, 500 + c, invalid_tones=(c % 5 == 0)) if c % 3 else S.random_stream(n_fr, 500 + c)
                   for c in range(n_ch)])
    pcm, rv, state = api.codec_decode_batch(fr)
    total = 0
    for c in range(n_ch):
        want, wrv = oracle_lib.ambe_decode(fr[c])
        assert (rv[c] == wrv).all()
        total += int((pcm[c] != want).sum())
    print("many channels: %d of %d samples differ" % (total, pcm.size))
    assert total == 0


def test_piecewise_decoding_carries_the_state(api):
    fr = np.stack([S.mixed_stream(120, 900 + c) for c in range(8)])
    whole, rv, _ = api.codec_decode_batch(fr)
    state = None
    parts = []
    for a, b in ((0, 1), (1, 40), (40, 41), (41, 120)):
        p, _, state = api.codec_decode_batch(fr[:, a:b], state=state)
        parts.append(p)
    assert np.array_equal(np.concatenate(parts, axis=1), whole)


def test_reference_style_calls(api):
    fr = S.mixed_stream(60, 42)
    want, wrv = oracle_lib.ambe_decode(fr)
    c = api.Codec()
    for i, f in enumerate(fr):
        audio, rc = c.decode_frame(f)
        assert rc == wrv[i]
        assert np.array_equal(audio[:160], want[i])
    # a tone frame over N = 80 and over N = 400 samples; dtx leaves the decoder alone
    d = oracle_lib.AmbeDecoder()
    d.decode(fr)
    for N in (80, 400):
        t = S.tone_frame(0x93, 240, sel=3)
        audio, rc = c.decode_frame(t, N=N)
        w, wrc = d.decode_frame(t, N=N)
        assert rc == wrc == 0 and np.array_equal(audio[:N], w[:N]) and audio[:N].any()
    z, rc = c.decode_dtx(50)
    assert rc == 0 and not z.any()
    audio, rc = c.decode_frame(fr[3])
    w, _ = d.decode_frame(fr[3])
    assert rc == 0 and np.array_equal(audio[:160], w[:160])
    # an unassigned tone code
    audio, rc = c.decode_frame(S.tone_frame(0x7f, 200, sel=3))
    assert rc == -22
    c.release()


def test_first_frame_with_interpolation_follows_the_oracle(api):
    """Decision D10: where the reference would run off its arrays the band edges are cut at the last bin."""
    for rule in (1, 2, 3):
        fr = S.speech_like(25, 60 + rule)
        fr[0, 6] = (fr[0, 6] & 0x3f) | (rule << 6)
        pcm, rv, _ = api.codec_decode_batch(fr[None])
        want, _ = oracle_lib.ambe_decode(fr)
        compare(pcm[0], want, "first frame, rule %d" % rule)


def test_device_pointers_and_streams(api):
    import torch
    dev = torch.device("cuda:0")
    n_ch, n_fr = 16, 50
    fr = np.stack([S.speech_like(n_fr, 300 + c) for c in range(n_ch)])
    d_fr = torch.from_numpy(fr).to(dev)
    d_pcm = torch.zeros((n_ch, n_fr, 160), dtype=torch.int16, device=dev)
    d_rv = torch.full((n_ch, n_fr), 7, dtype=torch.int32, device=dev)
    d_st = torch.zeros((n_ch, api.codec_state_bytes()), dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        api.codec_init_dev(st.cuda_stream, n_ch, d_st.data_ptr())
        api.codec_decode_batch_dev(st.cuda_stream, n_ch, n_fr, d_fr.data_ptr(), d_pcm.data_ptr(), d_rv.data_ptr(), d_st.data_ptr())
    st.synchronize()
    host, hrv, _ = api.codec_decode_batch(fr)
    assert np.array_equal(d_pcm.cpu().numpy(), host) and not d_rv.cpu().numpy().any()


def test_the_reference_program_running_on_this_library(gold):
    """src/gmr1_ambe_decode.c, unchanged, linked against libgmr1_hip.so (built in the container, where the reference's
    sources are: oracle/_ref/gmr1_ambe_decode_hip) reads a file of frames and writes the file the reference writes.
    It is a second GPU process started from this one, so it only runs on request and on its own, before this process
    has touched the GPU:  GMR1_RUN_REF_PROGRAM=1 python -m pytest tests/test_gpu_ambe.py -m gpu -k running_on_this_library
    (profiles/archive/r02r_ref_program_on_hip.log is such a run)."""
    import ref_codec
    if os.environ.get("GMR1_RUN_REF_PROGRAM") != "1":
        pytest.skip("set GMR1_RUN_REF_PROGRAM=1 and select this test alone")
    exe = ref_codec.build_program_on_product()
    if not exe:
        pytest.skip("the program was not built (no reference sources on the machine that made this snapshot)")
    fr = gold["mixed_frames"]
    got = ref_codec.decode_with_program(fr, tool=exe)
    assert got.shape == gold["mixed_pcm"].shape
    compare(got, gold["mixed_pcm"], "reference main() on libgmr1_hip.so")
    wav = ref_codec.decode_with_program(fr[:50], tool=exe, wav=True)
    assert np.array_equal(wav, got[:50])


def test_traffic_channel_to_pcm_end_to_end(api, orc, pkg):
    """What a user of the reference does in two programs - gmr1_rx writes the speech frames of a TCH3 assignment to a
    file (gmr1_rx.c:560-577), gmr1_ambe_decode turns the file into audio - as two GPU calls: samples of a BCCH carrier and
    its traffic carrier -> records (gmr1_hip_rx_run_tch) -> the 20 bytes of every speech burst, in order -> PCM.
    Checked against the same chain on the CPU oracle."""
    import workloads
    b, t, _, sent_t = workloads.bcch_tch_pair(pkg, 21, seconds=6.0, mix=(0.15, 0.75, 0.10))
    length = np.array([b.size], np.uint64)
    offset = np.zeros(1, np.uint64)
    rec, status, chains, found = api.rx_run_tch(b, t, offset, length, sps=4)
    assert not status.any()
    speech = rec[rec["type"] == 0x10]
    assert len(speech) > 60
    frames = np.stack([r["l2"][:20] for r in speech]).reshape(-1, 10)        # frame 0, frame 1 of each burst
    pcm, rv, _ = api.codec_decode_batch(frames[None])
    # the oracle's chain
    orv, orec, _ = orc.rx_run_tch(b, t, sps=4, arfcn=0)
    ospeech = orec[orec["type"] == 0x10]
    oframes = np.stack([r["l2"][:20] for r in ospeech]).reshape(-1, 10)
    assert np.array_equal(frames, oframes)
    want, wrv = oracle_lib.ambe_decode(oframes)
    assert np.array_equal(rv[0], wrv)
    compare(pcm[0], want, "TCH3 bursts -> PCM")
    # and the frames are the ones that were sent
    sent = [bytes(s["frame0"]) + bytes(s["frame1"]) for s in sent_t if s["type"] == "speech" and not s["ciph"]]
    got = {bytes(r["l2"][:20]) for r in speech}
    assert sum(s in got for s in sent) >= 0.9 * len(sent)


def test_bench_sized_batch_properties(api, pkg):
    """At the bench's channel count: channels do not influence each other (a permutation of the channels permutes the
    output), a run repeats bit for bit, and a random sample of channels equals the oracle."""
    n_ch, n_fr = 8192, 24
    fr = pkg.synth.ambe_speech_frames(n_ch, n_fr, seed=31)
    # sprinkle silence and tone frames
    rng = np.random.default_rng(32)
    for c in rng.integers(0, n_ch, 600):
        f = int(rng.integers(1, n_fr))
        fr[c, f] = S.silence_frame() if rng.random() < 0.5 else S.tone_frame(int(rng.integers(0x80, 0xa4)), 230, int(rng.integers(0, 4)))
    pcm, rv, st = api.codec_decode_batch(fr)
    pcm2, rv2, st2 = api.codec_decode_batch(fr)
    assert np.array_equal(pcm, pcm2) and np.array_equal(st, st2) and not rv.any()
    perm = rng.permutation(n_ch)
    pcm3, _, st3 = api.codec_decode_batch(fr[perm])
    assert np.array_equal(pcm3, pcm[perm]) and np.array_equal(st3, st[perm])
    differ = 0
    for c in rng.integers(0, n_ch, 48):
        want, _ = oracle_lib.ambe_decode(fr[c])
        differ += int((pcm[c] != want).sum())
    print("bench-sized batch: %d of %d sampled samples differ from the oracle" % (differ, 48 * n_fr * 160))
    assert differ == 0
