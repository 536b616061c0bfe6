#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mbursts/s of fused pi/4-CQPSK demod + K=5 Viterbi.

Workload (BASELINE.json configs[2], the one the metric is quoted on): a batch of
100 000 synthetic normal bursts per GPU, BCCH:CCCH = 1:6, windows of 1016 / 976
complex64 samples at sps=4, TOA jitter +-8 samples + fractional, CFO N(0, 50 Hz),
random phase / gain, Es/N0 in {6, 10, 20} dB.  One "step" = one pass of the hot
path (gmr1_hip_rx_bcch_ccch_batch_dev) over the whole batch, inputs resident in
HBM.  With N GPUs every rank processes its own batch (bursts are independent:
weak scaling, no data-path collective).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BYTES_BCCH = 1016 * 8 + 24 + 16     # window IQ + L2 + (crc, conv, toa, freq_err)
BYTES_CCCH = 976 * 8 + 24 + 16
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)


def kernel_sources_hash():
    """sha256 over the sources the headline kernel is compiled from: profiles/hbm_traffic.json names the build its
    PMC counters were taken on, and a traffic figure from another build is not reported."""
    import hashlib
    h = hashlib.sha256()
    for f in ("rx_kernels.hip", "tch3_body.h", "gmr1_dev.h", "rx_loop.h"):
        with open(os.path.join(ROOT, "osmo-gmr_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--preroll-s", type=float, default=0.3,
                    help="untimed seconds of the same step before the warm-up steps: brings the GPU out of its idle clocks "
                         "(a cold launch of this 0.3 ms kernel runs about 20 %% slower); 0 disables")
    ap.add_argument("--bursts", type=int, default=100_000, help="bursts per GPU per step")
    ap.add_argument("--cpu-sample", type=int, default=100_000, help="bursts timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--cpu-passes", type=int, default=3,
                    help="bursts workload: passes of the CPU oracle over its sample (3 x 100k bursts = about 11 s of CPU work)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--nt3-two-launches", action="store_true",
                    help="nt3 workload: demodulate and decode the speech bursts with the two separate calls")
    ap.add_argument("--workload", default="bursts", choices=["bursts", "fcch", "tch3", "nt3", "rx", "chan", "ambe"],
                    help="bursts = configs[2] (default, the headline metric); fcch = configs[1] rough sweep "
                         "over 1-s streams; tch3 = configs[4] l1-only TCH3 decode; nt3 = configs[4] from samples (90 %% speech + 10 %% FACCH3: demod + "
                         "layer 1); rx = configs[3] the whole "
                         "gmr1_rx loop (FCCH acquisition + BCCH/CCCH frame loop) over a multi-ARFCN capture")
    ap.add_argument("--wide-seconds", type=float, default=20.0, help="chan workload: wideband capture length at 2.0 Msps")
    ap.add_argument("--arfcns", type=int, default=64, help="rx workload: BCCH carriers per GPU")
    ap.add_argument("--seconds", type=float, default=60.0, help="rx workload: capture length")
    ap.add_argument("--streams", type=int, default=1024, help="fcch workload: 1-s streams per GPU")
    ap.add_argument("--channels", type=int, default=8192, help="ambe workload: voice channels per GPU")
    ap.add_argument("--frames", type=int, default=100, help="ambe workload: 20 ms frames per channel and step")
    ap.add_argument("--shard-arfcns", type=int, default=64, help="N > 1: carriers of the sharded config-4 run (extra keys)")
    ap.add_argument("--shard-seconds", type=float, default=60.0, help="N > 1: capture length of the sharded config-4 run")
    ap.add_argument("--no-shard", action="store_true", help="N > 1: skip the sharded config-4 run")
    ap.add_argument("--shard-timeout", type=float, default=240.0,
                    help="N > 1: if the sharded config-4 run has not finished after this many seconds, the headline line is "
                         "printed without it (with the reason) instead of hanging the job")
    ap.add_argument("--conv-decoder", default="acc", choices=["generic", "acc"],
                    help="which libosmocore Viterbi decoder the layer-1 chains reproduce (gmr1_hip_set_conv_decoder): "
                         "osmo_conv_decode_acc (default, as in the library: what every libosmocore since 2017 runs for these "
                         "codes) or its generic one; the CPU oracle beside it runs the same one")
    ap.add_argument("--layout", default="interleaved", choices=["interleaved", "planar"],
                    help="bursts workload: sample layout the TIMED step reads. interleaved (default) is the layout BASELINE "
                         "describes and the one `value` is quoted on; the default run also times the opt-in polyphase-planar "
                         "entry point afterwards and reports it as `roofline_planar`. --layout planar makes the planar call the "
                         "timed step itself (a side measurement for the profiler: config.layout says so)")
    ap.add_argument("--side-runner", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-extras", action="store_true",
                    help="bursts workload: skip the measurements after the timed region (planar layout, the other decoder)")
    args = ap.parse_args()
    # read by the library on first use and by tests/oracle_lib.py when it loads the oracle; inherited by spawned ranks
    os.environ["GMR1_HIP_CONV_DECODER"] = args.conv_decoder
    os.environ["ORC_CONV_MODE"] = "1" if args.conv_decoder == "acc" else "0"
    return args


def valu_profile(name):
    """profiles/valu_<name>.json (tools/valu_summary.py: machine-wide vector-ALU busy share and resident waves of a kernel from
    the builder's own PMC run) -- reported only while the kernel sources it was taken on are the ones in the tree."""
    vf = os.path.join(ROOT, "profiles", f"valu_{name}.json")
    if not os.path.exists(vf):
        return None
    try:
        vj = json.load(open(vf))
        if vj.get("kernel_sources_sha256") != kernel_sources_hash():
            return {"note": f"profiles/valu_{name}.json was taken on other kernel sources: not reported"}
        return {"kernel": vj.get("kernel"), "valu_busy": vj["valu_busy"], "avg_resident_waves": vj["avg_resident_waves"],
                "compiled_waves_per_simd": vj.get("compiled_waves_per_simd"), "valu_insts_per_wave": vj.get("valu_insts_per_wave"),
                "source": f"builder-measured, hash-gated file: {vj.get('source')} (build {vj.get('tag')}); machine-wide: sum SQ_ACTIVE_INST_VALU "
                          "resp. sum SQ_WAVE_CYCLES / (1024 SIMDs x kernel quad-cycles)"}
    except Exception as e:
        return {"note": f"profiles/valu_{name}.json unreadable: {e!r}"}


def emit(out, **kw):
    """Print the one JSON line; every line names the Viterbi decoder the run reproduced."""
    if isinstance(out.get("config"), dict):
        out["config"]["conv_decoder"] = os.environ.get("GMR1_HIP_CONV_DECODER", "acc")
        # which build answered: the product library unless a profiling script pointed GMR1_HIP_LIBRARY elsewhere
        lib = os.environ.get("GMR1_HIP_LIBRARY")
        out["config"]["library"] = os.path.basename(lib) if lib else "libgmr1_hip.so"
        if lib and "prof" in os.path.basename(lib):
            out["config"]["profiling_build"] = True
    print(json.dumps(out), **kw)


def preroll(step, seconds):
    """Run `step` for `seconds` of wall time, untimed, before the W warm-up steps: the GPU leaves its idle power state
    only under sustained load, and a measurement of K short launches right after an idle period sees the ramp, not the
    kernel.  Nothing is skipped or cached by this: it is the same step on the same data."""
    import torch
    # the first timing event recorded on a stream makes the runtime re-arm the queue for timestamps, and the dispatch
    # that follows waits for it (tens of milliseconds, once per process): that belongs to no step
    arm = torch.cuda.Event(enable_timing=True)
    arm.record(torch.cuda.current_stream())
    step()
    torch.cuda.synchronize()
    if seconds <= 0:
        return 0
    t_end = time.perf_counter() + seconds
    n = 0
    while time.perf_counter() < t_end:
        for _ in range(8):
            step()
        n += 8
        torch.cuda.synchronize()
    return n


def run_chan_workload(args):
    """BASELINE.md config 4, wideband container: a 2.0 Msps capture -> all 64 ARFCN streams at 93.6 ksps
    (gmr1_hip_channelize_dev: polyphase filterbank + per-channel root-raised-cosine resampler)."""
    import torch
    from __graft_entry__ import load_package
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    pkg = load_package()
    api = pkg.api
    api.load()
    api.init(0)
    fs = 2.0e6
    n_in = int(args.wide_seconds * fs) // 64 * 64
    n_chans, n_mid, n_out = api.channelize_plan(fs, 4, n_in)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    wide = torch.randn((n_in, 2), generator=g, device=dev, dtype=torch.float32)
    out = torch.empty((n_chans, n_out, 2), device=dev, dtype=torch.float32)
    chans = list(range(n_chans))
    stream = torch.cuda.current_stream(dev)

    def step():
        api.channelize_dev(stream.cuda_stream, wide.data_ptr(), n_in, fs, chans, out.data_ptr(), n_out)
    preroll(step, args.preroll_s)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / args.steps
    # algorithmic bytes: wideband read once, 2x oversampled channel streams written and read once, outputs written
    alg = n_in * 8 + 2 * n_chans * n_mid * 8 + n_chans * n_out * 8
    achieved = alg / (kern_ms * 1e-3) / 1e9
    outj = {"metric": "Mbursts/s demod+Viterbi (and IQ Msamp/s), 1/2/4/8 MI355X", "value": n_in * args.steps / wall / 1e6,
            "unit": "Msamp/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"wideband channelizer: {args.wide_seconds:g} s @ 2.0 Msps -> {n_chans} ARFCN streams @ 93.6 ksps "
                                   "(617-tap polyphase filterbank 2x oversampled + 32-phase RRC resampler)",
                       "realtime_factor": args.wide_seconds * args.steps / wall},
            "roofline": {"bound": "hbm", "kernel": "k_pfb64 + k_resamp", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": alg,
                         "limited_by": "hbm (both kernels' loads and stores alone take 85 % of their time; the resampler's reads and writes together run at the box's mixed rate and its arithmetic hides under neither completely: DESIGN 4.7)"}}
    if not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import orc_chan
        pl = orc_chan.Plan(fs)
        m = 400000
        x = wide[:m].cpu().numpy().view(np.complex64).reshape(-1)
        tc = time.perf_counter()
        ref = orc_chan.channelize(x, pl, [0, 21, 63])
        tc = time.perf_counter() - tc
        got = api.channelize(x, fs, [0, 21, 63])
        err = max(float(np.max(np.abs(got[i] - ref[k]))) for i, k in enumerate([0, 21, 63]))
        outj["cpu_baseline"] = {"value": m / tc / 1e6, "unit": "Msamp/s", "cores": 1, "kind": "port",
                                "sample": f"first {m} wideband samples, 3 of 64 output branches, numpy oracle, {tc:.1f} s"}
        outj["checks"] = {"max_abs_err_vs_oracle": err}
    emit(outj)


def run_ambe_workload(args):
    """SURVEY.md 8 row f4, last part: the AMBE vocoder (reference src/codec, gmr1_ambe_decode).  One step = `frames`
    consecutive 20 ms frames of each of `channels` voice channels -> PCM, decoder states carried in HBM from step to
    step like a running call (gmr1_hip_codec_decode_batch_dev)."""
    import torch
    from __graft_entry__ import load_package
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    pkg = load_package()
    api = pkg.api
    api.load()
    api.init(0)
    n_ch, n_fr = args.channels, args.frames
    t_gen = time.perf_counter()
    host = pkg.synth.ambe_speech_frames(n_ch, n_fr, seed=9)
    t_gen = time.perf_counter() - t_gen
    frames = torch.from_numpy(host).to(dev)
    pcm = torch.empty((n_ch, n_fr, 160), dtype=torch.int16, device=dev)
    rv = torch.empty((n_ch, n_fr), dtype=torch.int32, device=dev)
    state = torch.zeros((n_ch, api.codec_state_bytes()), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)
    api.codec_init_dev(stream.cuda_stream, n_ch, state.data_ptr())
    # the first step from fresh decoders is the one that is checked
    api.codec_decode_batch_dev(stream.cuda_stream, n_ch, n_fr, frames.data_ptr(), pcm.data_ptr(), rv.data_ptr(), state.data_ptr())
    torch.cuda.synchronize()
    first = pcm[:16].cpu().numpy()

    def step():
        api.codec_decode_batch_dev(stream.cuda_stream, n_ch, n_fr, frames.data_ptr(), pcm.data_ptr(), rv.data_ptr(),
                                   state.data_ptr())
    preroll(step, args.preroll_s)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / args.steps
    total = n_ch * n_fr
    # algorithmic bytes: 10 in + 320 out per frame, the decoder state read and written once per channel and launch
    alg = total * (10 + 320 + 4) + 2 * n_ch * api.codec_state_bytes()
    achieved = alg / (kern_ms * 1e-3) / 1e9
    outj = {"metric": "Mbursts/s demod+Viterbi (and IQ Msamp/s), 1/2/4/8 MI355X", "value": total * args.steps / wall / 1e6,
            "unit": "Mframes/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"AMBE speech decoder: {n_ch} voice channels x {n_fr} frames of 20 ms per step, "
                                   "10-byte frames -> 160 samples of 8 kHz PCM, decoder states resident",
                       "realtime_factor": total * 0.02 * args.steps / wall, "pcm_msamp_per_s": total * 160 * args.steps / wall / 1e6},
            "roofline": {"bound": "hbm", "kernel": "k_ambe", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": alg, "limited_by": "valu_issue + dependent chains",
                         "note": "330 bytes per frame against tens of thousands of table-cosine multiply-adds: the kernel is "
                                 "VALU / LDS bound by construction (DESIGN.md 4.7), the HBM fraction is reported for form"},
            "checks": {"workload_gen_s": round(t_gen, 2), "rejected_frames": int((rv != 0).sum().item())}}
    if not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        import ref_codec
        m_ch = min(n_ch, 2048)
        have_ref = os.path.exists(ref_codec.LIB)
        tc = time.perf_counter()
        differ = 0
        for c in range(m_ch):
            if have_ref:
                got, _ = ref_codec.decode_clean_stack(host[c])
            else:
                got, _ = oracle_lib.ambe_decode(host[c])
            if c < 16 and not have_ref:
                differ += int((got != first[c]).sum())
        tc = time.perf_counter() - tc
        if have_ref:      # the check is against the oracle's default reading (what the reference's program computes)
            differ = sum(int((oracle_lib.ambe_decode(host[c])[0] != first[c]).sum()) for c in range(16))
        outj["cpu_baseline"] = {"value": m_ch * n_fr / tc / 1e6, "unit": "Mframes/s", "cores": 1,
                                "kind": "reference" if have_ref else "port",
                                "sample": f"the first {m_ch} channels x {n_fr} frames of the same workload, "
                                          + ("the reference's src/codec compiled from its sources (oracle/_ref), one call "
                                             "per frame through oracle/ref_codec_shim.c" if have_ref else "gcc -O2 oracle")
                                          + f", 1 thread, {tc:.1f} s"}
        outj["checks"]["samples_differing_from_oracle_first_16_channels"] = differ
        outj["checks"]["samples_compared"] = int(first.size)
    emit(outj)


def run_rx_workload(args):
    """configs[3]: the reference's gmr1_rx loop over a channelised capture of
    --arfcns BCCH carriers x --seconds, device-resident, one gmr1_hip_rx_run_dev call per step."""
    import torch
    from __graft_entry__ import load_package
    import workloads
    import oracle_lib
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    pkg = load_package()
    api = pkg.api
    api.load()
    api.init(0)
    A, sps = args.arfcns, 4
    ns = int(args.seconds * 23400 * sps)
    distinct = min(A, 8)
    host = [workloads.bcch_carrier(pkg, 700 + a, seconds=args.seconds, sps=sps, stn=(5 * a) % 24, delay=a % 8,
                                   cfo_hz=40.0 * (a - 3), esn0_db=10.0 + a)[0] for a in range(distinct)]
    base = torch.from_numpy(np.concatenate(host).view(np.float32)).to(dev)
    iq = torch.cat([base] * (-(-A // distinct)))[:A * ns * 2].contiguous()
    offset = np.arange(A, dtype=np.uint64) * np.uint64(ns)
    length = np.full(A, ns, np.uint64)
    stream = torch.cuda.current_stream(dev)
    res = [None]
    # the caller's record buffer, reused across steps: pinned host memory, which the library copies the records into
    # directly (a pageable buffer costs one more host copy of the records, include/gmr1_hip.h)
    rec_pin = torch.empty((1 << 18) * api.RX_RECORD.itemsize, dtype=torch.uint8, pin_memory=True)
    rec_buf = rec_pin.numpy().view(api.RX_RECORD)

    # (arguments marshalled once, as a C host has them: the step is the library call)
    call = api.rx_run_dev_prepared(stream.cuda_stream, iq.data_ptr(), offset, length, rec_buf, sps=sps)

    def step():
        res[0] = call()
    preroll(step, args.preroll_s)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    rec, status, chains, found = res[0]
    # the library's own clock around its phases: the median over a few steps more, behind the timed region
    samples = [api.rx_run_last_timing()]
    for _ in range(min(10, args.steps)):
        step()
        samples.append(api.rx_run_last_timing())
    phases = {k: float(np.median([p[k] for p in samples])) for k in samples[0]}
    out = {"metric": "Mbursts/s demod+Viterbi (and IQ Msamp/s), 1/2/4/8 MI355X", "value": A * ns * args.steps / wall / 1e6,
           "unit": "Msamp/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"configs[3]: gmr1_rx loop, {A} BCCH carriers x {args.seconds:g} s @ 93.6 ksps "
                                  f"({distinct} distinct, tiled), FCCH acquisition + BCCH/CCCH frame loop with feedback",
                      "frames_decoded": int(found), "chains": int(chains.sum()),
                      "realtime_factor": A * args.seconds * args.steps / wall},
           "roofline": None}
    # The loop is a chain of dependent bursts per carrier (the next BCCH window is placed by the previous one's timing
    # and frequency): latency-bound by construction.  The figure below prices only what it has to read -- the windows of
    # the frames it decoded -- over the whole step (acquisition + k_rx_chain / k_rx4 / k_rx_merge + record collection), to show how far from
    # the HBM roofline a feedback loop sits; it is not a kernel-quality number.
    alg = float(found) * 7893.7
    step_s = wall / args.steps
    out["roofline"] = {"bound": "hbm", "kernel": "k_rx_chain + k_rx4 + k_rx_merge (+ FCCH acquisition), whole step", "achieved": alg / step_s / 1e9,
                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / step_s / 1e9 / HBM_PEAK_GBS, "traffic": None,
                       "kernel_ms": step_s * 1e3, "algorithmic_bytes_per_launch": alg,
                       "limited_by": "latency (dependent chain)",
                       "note": "latency-bound feedback chain: 187 dependent BCCH bursts per carrier-minute, each waiting for the "
                               "front half of the one before (DESIGN.md 4.4); the HBM fraction is nominal for it"}
    out["phases_ms"] = {k: round(v, 4) for k, v in phases.items()}
    if not args.no_cpu:
        oracle_lib.lib()
        tc = time.perf_counter()
        orv, orec, och = oracle_lib.rx_run(host[0], sps=sps, arfcn=0)
        tc = time.perf_counter() - tc
        mine = rec[rec["arfcn"] == 0]
        key = lambda r: [(int(x["chain"]), int(x["type"]), int(x["fn"]), int(x["tn"]), bytes(x["l2"])) for x in r]
        out["cpu_baseline"] = {"value": ns / tc / 1e6, "unit": "Msamp/s", "cores": 1, "kind": "port",
                               "sample": f"carrier 0 ({args.seconds:g} s), gcc -O2 oracle, 1 thread, {tc * 1e3:.0f} ms"}
        out["checks"] = {"frames_identical_to_oracle": bool(key(mine) == key(orec)), "oracle_frames": int(len(orec))}
        # the same port on every host core: the A carriers dealt out to one thread per core (the reference would run one
        # gmr1_rx process per carrier file)
        from concurrent.futures import ThreadPoolExecutor
        cores = os.cpu_count() or 1

        def work(t):
            for a_ in range(t, A, cores):
                oracle_lib.rx_run(host[a_ % distinct], sps=sps, arfcn=a_)
        ta = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(work, range(cores)))
        ta = time.perf_counter() - ta
        out["cpu_baseline_all_cores"] = {"value": A * ns / ta / 1e6, "unit": "Msamp/s", "cores": cores, "kind": "port",
                                         "sample": f"all {A} carriers, one oracle loop per carrier, {cores} threads, {ta * 1e3:.0f} ms"}
        out["gpu_vs_all_cores"] = out["value"] / out["cpu_baseline_all_cores"]["value"]
    # What 8 GPUs would do with THIS capture (strong scaling): every rank gets A / 8 carriers.  The loop is a chain of
    # dependent rounds per carrier (rx_bcch feeds align and a FLOAT freq_err back, gmr1_rx.c:782-789), so fewer carriers per
    # GPU shorten nothing but the acquisition: measured here by running A / 8 carriers on this one GPU; the scatter of the
    # other ranks' samples from rank 0 is priced at the per-link xGMI rate (7 peers at once, one link each).
    if A >= 8 and A % 8 == 0:
        A8 = A // 8
        res8 = [None]

        def step8():
            res8[0] = api.rx_run_dev(stream.cuda_stream, iq.data_ptr(), offset[:A8], length[:A8], sps=sps, out=rec_buf)
        for _ in range(3):
            step8()
        torch.cuda.synchronize()
        t8 = time.perf_counter()
        for _ in range(args.steps):
            step8()
        torch.cuda.synchronize()
        t8 = (time.perf_counter() - t8) / args.steps
        scatter_s = A8 * ns * 8 / 153e9
        out["projected_n8"] = {"carriers_per_gpu": A8, "rx_ms_per_gpu_measured_here": t8 * 1e3, "scatter_ms_at_153_GB_per_s_per_link": scatter_s * 1e3,
                               "total_ms": (t8 + scatter_s) * 1e3, "speedup_over_one_gpu": step_s / (t8 + scatter_s),
                               "speedup_without_scatter": step_s / t8,
                               # the floor of ANY root scatter: 7/8 of the capture has to leave rank 0 over its seven links, so a
                               # scatter chunked in time with the walk gated on arrival flags (not built) could at best hide the
                               # loop under the transfer -- it cannot bring the exchange under the transfer's own time
                               "total_ms_if_the_loop_ran_under_the_scatter": (max(t8, scatter_s) + scatter_s / max(args.seconds, 1.0)) * 1e3,
                               "note": "latency chain: the rounds of a carrier cannot be shortened by adding GPUs; 8 GPUs carry 8x the carriers in the same time (weak scaling), they do not finish these sooner"}
    emit(out)


def run_side_workload(args):
    """Secondary workloads (configs[1] FCCH sweep, configs[4] TCH3 l1-only): same timing contract,
    single GPU, used for the DESIGN.md tables.  The default run never comes here."""
    import torch
    from __graft_entry__ import load_package
    import workloads
    import oracle_lib
    rank, world, backend, dev, dev_index, grouped = init_ranks()     # (only tch3 comes here with more than one rank)
    pkg = load_package()
    api = pkg.api
    api.load()
    api.init(dev_index)
    stream = torch.cuda.current_stream(dev)
    n_global = None
    if args.workload == "fcch":
        n = args.streams
        wl = workloads.fcch_streams(pkg, n, seed=2)
        ns = wl["n_samples"]
        iq = torch.from_numpy(wl["iq"].view(np.float32)).to(dev)
        offset = torch.from_numpy(wl["offset"].astype(np.int64)).to(dev)
        toa = torch.zeros(n, dtype=torch.int32, device=dev)
        rv = torch.zeros(n, dtype=torch.int32, device=dev)

        import ctypes as C
        f_fine = api.load().gmr1_hip_fcch_fine_batch_dev
        f_fine.restype = C.c_int
        ftoa = torch.zeros(n, dtype=torch.int32, device=dev)
        ferr = torch.zeros(n, dtype=torch.float32, device=dev)

        def step():
            api.fcch_rough_batch_dev(stream.cuda_stream, "fcch", n, 4, ns, iq.data_ptr(), offset.data_ptr(), None,
                                     toa.data_ptr(), rv.data_ptr())
            # fine stage on the burst the rough sweep found (fcch_single_init, gmr1_rx.c:605-639); the offsets
            # stay on the device
            off_f = offset + torch.clamp(toa.to(torch.int64), 0, ns - 117 * 4)
            rc = f_fine(C.c_void_p(stream.cuda_stream), C.c_int(0), C.c_int(n), C.c_int(4), C.c_void_p(iq.data_ptr()),
                        C.c_void_p(off_f.data_ptr()), None, C.c_void_p(ftoa.data_ptr()), C.c_void_p(ferr.data_ptr()))
            assert rc == 0
        units, unit = n * ns / 1e6, "Msamp/s"
        bytes_per_launch = n * (ns * 8 + 4)
        kernel = "k_fcch_sweep<117> + k_fcch_energy<117> + k_fcch_pick (+ k_fcch_fine)"
        workload = (f"configs[1]: FCCH rough + fine, {n} x 1-s streams @ 93.6 ksps (rough: 23 284 lags x 117 taps each; "
                    "fine: 117-point DFT of the found burst)")
    else:
        # N GPUs: ONE global workload, rank r decodes the contiguous block shard.partition_contiguous gives it (edges on
        # multiples of 4 as for the FACCH3 groups of the from-samples workload); strong scaling, no collective
        n_global = args.bursts * 10
        wl = workloads.tch3_bursts(pkg, n_global, seed=5)
        g0, g1 = pkg.shard.partition_contiguous(n_global, world, rank, group=4)
        n = g1 - g0
        eb_all = wl["ebits"]
        wl = dict(wl, ebits=eb_all[g0:g1])
        eb = torch.from_numpy(wl["ebits"]).to(dev)
        frames = torch.zeros((n, 2, 10), dtype=torch.uint8, device=dev)
        st = torch.zeros((n, 4), dtype=torch.uint8, device=dev)
        conv = torch.zeros((n, 2), dtype=torch.int32, device=dev)
        f = api.load().gmr1_hip_tch3_decode_batch_dev
        import ctypes as C

        def step():
            rc = f(C.c_void_p(stream.cuda_stream), C.c_int(n), C.c_int(0), C.c_void_p(eb.data_ptr()), None,
                   C.c_void_p(frames.data_ptr()), C.c_void_p(st.data_ptr()), C.c_void_p(conv.data_ptr()))
            assert rc == 0
        units, unit = n_global / 1e6, "Mbursts/s"
        bytes_per_launch = n * (212 + 24 + 8)
        kernel = "k_tch3"
        workload = (f"configs[4] l1-only: {n_global} NT3 speech bursts, descramble + 104-perm + punctured K=7 tail-biting Viterbi"
                    + (f"; contiguous blocks over {world} ranks, no collective" if world > 1 else ""))

    def barrier():
        if grouped:
            import torch.distributed as dist
            dist.barrier()
    preroll(step, args.preroll_s)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize()
    barrier()
    wall = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / args.steps
    sharded_same = None
    if grouped:
        import torch.distributed as dist
        tt = torch.tensor([wall], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())
        if world > 1:
            got = gather_bytes(frames.cpu().numpy(), rank, world, backend, dev)
            if rank == 0:
                # ONE run over the whole workload on this GPU: what the ranks' blocks, concatenated, must equal
                eb1 = torch.from_numpy(eb_all).to(dev)
                fr1 = torch.zeros((n_global, 2, 10), dtype=torch.uint8, device=dev)
                rc = f(C.c_void_p(stream.cuda_stream), C.c_int(n_global), C.c_int(0), C.c_void_p(eb1.data_ptr()), None,
                       C.c_void_p(fr1.data_ptr()), None, None)
                assert rc == 0
                torch.cuda.synchronize()
                sharded_same = bool(np.array_equal(got, fr1.cpu().numpy().reshape(-1)))
        if rank != 0:
            dist.destroy_process_group()
            return
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
    out = {"metric": "Mbursts/s demod+Viterbi (and IQ Msamp/s), 1/2/4/8 MI355X", "value": units * args.steps / wall,
           "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if world > 1 else "weak",
           "vs_baseline": None, "dtype": "f32" if args.workload == "fcch" else "i32", "data": "synthetic",
           "config": {"workload": workload},
           "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel_ms": kern_ms,
                        "algorithmic_bytes_per_launch": bytes_per_launch}}
    if args.workload == "tch3":
        # 244 algorithmic bytes per burst against ~ 825 vector instructions: the 8 TB/s roof is nominal for this kernel
        out["roofline"]["limited_by"] = "valu_issue"
        vp = valu_profile("k_tch3")
        if vp:
            out["roofline_valu"] = dict({"bound": "valu_issue"}, **vp)
    else:
        out["roofline"]["limited_by"] = "hbm"
    if args.workload == "fcch":
        # The sweep is a sliding correlation: every lag of every stream takes 117 multiply-adds of a complex sample with a
        # REAL tap (the dual chirp is real: 2 FMA = 4 flops executed; the reference's generic complex correlation spends 8)
        # on the FP32 vector pipe -- no contraction shape for MFMA (one stream against 117 taps at 23 284 lags).  Besides the HBM
        # fraction of the whole step above: the rough sweep alone (k_fcch_stats + k_fcch_corr + k_fcch_pick, timed here
        # after the timed region) against the FP32 vector peak.
        def rough_only():
            api.fcch_rough_batch_dev(stream.cuda_stream, "fcch", n, 4, ns, iq.data_ptr(), offset.data_ptr(), None,
                                     toa.data_ptr(), rv.data_ptr())
        for _ in range(5):
            rough_only()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(args.steps):
            rough_only()
        e1.record(stream)
        torch.cuda.synchronize()
        rough_ms = e0.elapsed_time(e1) / args.steps
        nlags = ns // 4 - 117 + 1
        flops = float(n) * nlags * 117 * 4
        peak_tf = 157.3                      # MI355X FP32 vector peak (256 CUs x 128 lanes x 2 flop x 2.4 GHz)
        out["roofline_valu"] = {"bound": "fp32 (matrix peak = vector peak on this part)", "kernel": "k_fcch_sweep<117> + k_fcch_energy<117> + k_fcch_pick (rough sweep alone)",
                                "achieved": flops / (rough_ms * 1e-3) / 1e12, "peak": peak_tf, "unit": "TFLOP/s",
                                "frac": flops / (rough_ms * 1e-3) / 1e12 / peak_tf, "kernel_ms": rough_ms,
                                "algorithmic_flops_per_launch": flops,
                                "note": "useful flops (real taps; the banded Toeplitz form on the matrix cores executes 132 / 117 of them); "
                                        "the sweep kernel streams the window from HBM and correlates in the same launch, so this time is the "
                                        "HBM-bound one: the arithmetic is no longer what the step waits for"}
    if sharded_same is not None:
        out["checks"] = {"sharded_outputs_identical_to_single_gpu_run": sharded_same}
    # CPU baseline + parity on a bounded sample
    if not args.no_cpu and world == 1:
        oracle_lib.lib()
        if args.workload == "fcch":
            m = min(n, 24)
            tc = time.perf_counter()
            ref = [oracle_lib.fcch_rough(wl["iq"][i], 4)[1] for i in range(m)]
            tc = time.perf_counter() - tc
            got = toa.cpu().numpy()[:m]
            out["cpu_baseline"] = {"value": m * ns / tc / 1e6, "unit": "Msamp/s", "cores": 1, "kind": "port",
                                   "sample": f"first {m} streams, gcc -O2 oracle, 1 thread, {tc:.1f} s"}
            out["checks"] = {"toa_identical_to_oracle": bool(np.array_equal(got, np.array(ref)))}
        else:
            m = min(n, 20000)
            tc = time.perf_counter()
            ref = oracle_lib.tch3_decode(wl["ebits"][:m], 0)
            tc = time.perf_counter() - tc
            g0 = frames.cpu().numpy()[:m]
            out["cpu_baseline"] = {"value": m / tc / 1e6, "unit": "Mbursts/s", "cores": 1, "kind": "port",
                                   "sample": f"first {m} bursts, gcc -O2 oracle via ctypes (per-burst call), 1 thread, {tc:.1f} s"}
            out["checks"] = {"frames_identical_to_oracle": bool(np.array_equal(g0[:, 0], ref[0]) and
                                                                 np.array_equal(g0[:, 1], ref[1])),
                             "conv_identical": bool(np.array_equal(conv.cpu().numpy()[:m, 0], ref[3]))}
    emit(out, flush=True)
    if grouped:
        import torch.distributed as dist
        dist.destroy_process_group()


def run_nt3_workload(args):
    """BASELINE.md configs[4] from samples: NT3 bursts, 90 % speech and 10 % FACCH3 in
    groups of four, window 474 samples; per step: speech bursts from samples to speech frames (one launch), FACCH3 bursts
    demodulated and their groups decoded -- everything resident in HBM.  100 k distinct bursts are generated on the host
    (the same on every rank: one seed) and global burst g is distinct burst g mod 100 k, each in its own memory (the tiles
    are not cache hits of each other).
    N GPUs: ONE global workload of --bursts x 10 bursts, rank r takes the contiguous block
    shard.partition_contiguous(n, N, r, group=4) -- block edges on FACCH3 groups, which decode as a unit
    (facch3.c:121-158) -- no data-path collective, barrier + max-over-ranks timing, `scaling: strong`; after the timed
    region rank 0 gathers every rank's speech frames and FACCH3 results and compares them with ONE run over the whole
    workload on its own GPU."""
    import ctypes as C
    import torch
    from __graft_entry__ import load_package
    import workloads
    rank, world, backend, dev, dev_index, grouped = init_ranks()
    pkg = load_package()
    api = pkg.api
    api.load()
    api.init(dev_index)
    L = api.load()
    stream = torch.cuda.current_stream(dev)
    n = args.bursts * 10
    base = min(n, 100_000)
    base -= base % 40
    reps = max(1, n // base)
    n = base * reps                                    # the global workload
    t_gen = time.time()
    wl = workloads.nt3_mix(pkg, base, seed=5)
    t_gen = time.time() - t_gen
    stride = wl["stride"]
    base_dev = torch.from_numpy(wl["iq"].view(np.float32)).to(dev).view(base, stride * 2)
    P = lambda t: C.c_void_p(t.data_ptr())
    sp = C.c_void_p(stream.cuda_stream)
    id_f = api.BURST_IDS.index("nt3_facch")

    def resident(g0, g1):
        """Inputs and result buffers of the global bursts [g0, g1) in this rank's HBM."""
        gi = np.arange(g0, g1, dtype=np.int64)
        bi = gi % base
        W = {"n": g1 - g0, "iq": base_dev.index_select(0, torch.from_numpy(bi).to(dev)).reshape(-1)}
        fac = (gi % 40) >= 36
        loc = np.arange(g1 - g0, dtype=np.int64)
        W["off_s"] = torch.from_numpy(loc[~fac] * stride).to(dev)
        W["off_f"] = torch.from_numpy(loc[fac] * stride).to(dev)
        W["fs_s"] = torch.from_numpy(wl["freq_shift"][bi[~fac]]).to(dev)
        W["fs_f"] = torch.from_numpy(wl["freq_shift"][bi[fac]]).to(dev)
        n_s, n_f = W["off_s"].numel(), W["off_f"].numel()
        assert n_f % 4 == 0
        W["n_s"], W["n_f"] = n_s, n_f
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        W.update(eb_s=z((n_s, 212), torch.int8), eb_f=z((n_f, 104), torch.int8), sid_s=z(n_s, torch.int32), sid_f=z(n_f, torch.int32),
                 toa_s=z(n_s, torch.float32), toa_f=z(n_f, torch.float32), rv_s=z(n_s, torch.int32), rv_f=z(n_f, torch.int32),
                 frames=z((n_s, 2, 10), torch.uint8), st=z((n_s, 4), torch.uint8), conv_s=z((n_s, 2), torch.int32),
                 l2f=z((n_f // 4, 10), torch.uint8), bs_f=z((n_f // 4, 32), torch.uint8), crc_f=z(n_f // 4, torch.int32),
                 conv_f=z(n_f // 4, torch.int32))
        return W

    def run_step(W, want_eb=False):
        n_s, n_f = W["n_s"], W["n_f"]
        rc = 0
        if n_s and args.nt3_two_launches:
            id_s = api.BURST_IDS.index("nt3_speech")
            rc = L.gmr1_hip_demod_batch_dev(sp, C.c_int(id_s), C.c_int(n_s), C.c_int(4), C.c_int(474), P(W["iq"]), P(W["off_s"]),
                                            P(W["fs_s"]), P(W["eb_s"]), C.c_int(212), P(W["sid_s"]), P(W["toa_s"]), None, None, P(W["rv_s"]))
            rc |= L.gmr1_hip_tch3_decode_batch_dev(sp, C.c_int(n_s), C.c_int(0), P(W["eb_s"]), None, P(W["frames"]), P(W["st"]),
                                                   P(W["conv_s"]))
        elif n_s:
            # rx_tch3's burst step as one call (one launch: the soft bits stay in LDS and, in the timed steps, are not
            # written out -- rx_tch3 has no use for them either; the one step after the timed ones asks for them, for the checks)
            rc = L.gmr1_hip_tch3_rx_batch_dev(sp, C.c_int(n_s), C.c_int(4), C.c_int(474), P(W["iq"]), P(W["off_s"]), P(W["fs_s"]),
                                              C.c_int(0), None, P(W["eb_s"]) if want_eb else None, P(W["sid_s"]), P(W["toa_s"]),
                                              P(W["rv_s"]), P(W["frames"]), P(W["st"]), P(W["conv_s"]))
        if n_f:
            rc |= L.gmr1_hip_demod_batch_dev(sp, C.c_int(id_f), C.c_int(n_f), C.c_int(4), C.c_int(474), P(W["iq"]), P(W["off_f"]),
                                             P(W["fs_f"]), P(W["eb_f"]), C.c_int(104), P(W["sid_f"]), P(W["toa_f"]), None, None, P(W["rv_f"]))
            rc |= L.gmr1_hip_facch3_decode_batch_dev(sp, C.c_int(n_f // 4), P(W["eb_f"]), None, P(W["l2f"]), P(W["bs_f"]),
                                                     P(W["crc_f"]), P(W["conv_f"]))
        assert rc == 0, L.gmr1_hip_last_error()

    g0, g1 = pkg.shard.partition_contiguous(n, world, rank, group=4)
    W = resident(g0, g1)
    n_s, n_f = W["n_s"], W["n_f"]
    step = lambda: run_step(W)

    def barrier():
        if grouped:
            import torch.distributed as dist
            dist.barrier()

    preroll(step, args.preroll_s)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize()
    barrier()
    wall = time.perf_counter() - t0
    step_ms = ev0.elapsed_time(ev1) / args.steps
    if grouped:
        import torch.distributed as dist
        tt = torch.tensor([wall], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())
    run_step(W, want_eb=True)
    torch.cuda.synchronize()
    # ---- N > 1: every rank's results to rank 0, compared with one run over the whole workload on rank 0's GPU ----
    sharded_same = None
    if world > 1:
        got = [gather_bytes(W[k].cpu().numpy(), rank, world, backend, dev) for k in ("frames", "st", "l2f", "crc_f")]
        if rank == 0:
            Wall = resident(0, n)
            run_step(Wall)
            torch.cuda.synchronize()
            sharded_same = all(np.array_equal(g, Wall[k].cpu().numpy().reshape(-1).view(np.uint8))
                               for g, k in zip(got, ("frames", "st", "l2f", "crc_f")))
            del Wall
    if grouped:
        import torch.distributed as dist
        if rank != 0:
            dist.destroy_process_group()
            return
    # speech 474 x 8 + 20 + 4 + 12 = 3 828 B, FACCH3 474 x 8 per burst + (10 + 32 + 8) per group
    bytes_per_step = n_s * 3828 + n_f * 3792 + (n_f // 4) * 50
    achieved = bytes_per_step / (step_ms * 1e-3) / 1e9
    out = {"metric": "Mbursts/s demod+Viterbi (and IQ Msamp/s), 1/2/4/8 MI355X", "value": n * args.steps / wall / 1e6,
           "unit": "Mbursts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if world > 1 else "weak",
           "vs_baseline": None,
           "dtype": "f32+i32", "data": "synthetic",
           "config": {"workload": f"configs[4] from samples: {n} NT3 bursts ({n * 9 // 10} speech + {n // 10} FACCH3 in groups of 4, "
                                  f"{base} distinct, each in its own memory), window 474 @ sps 4: pi4cxpsk demod + TCH3 / FACCH3 layer 1",
                      "global_bursts": n, "bursts_on_rank_0": g1 - g0,
                      "parallelism": f"contiguous blocks of the one workload over {world} rank(s), edges on FACCH3 groups of 4, no collective"},
           "iq_msamp_per_s": n * 474 * args.steps / wall / 1e6,
           "roofline": {"bound": "hbm", "kernel": ("k_rx4g<8,4> + k_tch3" if args.nt3_two_launches else "k_rx4g_tch3") + " + k_rx4g<8,4,FAC> + k_facch3 (whole step, rank 0's block)", "achieved": achieved,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                        "kernel_ms": step_ms, "algorithmic_bytes_per_launch": bytes_per_step},
           "checks": {"workload_gen_s": round(t_gen, 1)}}
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic_nt3.json")
    if os.path.exists(tfile) and world == 1 and n == 1_000_000 and not args.nt3_two_launches:
        tj = json.load(open(tfile))
        if tj.get("kernel_sources_sha256") == kernel_sources_hash():
            out["roofline"]["traffic"] = tj.get("step_bytes_1M")
            out["roofline"]["traffic_source"] = (f"builder-measured, hash-gated file profiles/hbm_traffic_nt3.json (not measured in this run): PMC "
                                                 f"FETCH_SIZE x2 + WRITE_SIZE summed over the step's kernels, separate passes, build {tj.get('tag')}")
        else:
            out["roofline"]["traffic_source"] = "profiles/hbm_traffic_nt3.json was taken on other kernel sources: not reported"
    # nine tenths of the step are k_rx4g_tch3, whose vector ALUs are saturated (machine-wide busy share ~ 1.0): the limit is
    # instruction issue, the 8 TB/s roof is nominal
    out["roofline"]["limited_by"] = "valu_issue"
    vp = valu_profile("k_rx4g_tch3")
    if vp:
        out["roofline_valu"] = dict({"bound": "valu_issue"}, **vp)
    if sharded_same is not None:
        out["checks"]["sharded_outputs_identical_to_single_gpu_run"] = bool(sharded_same)
    # what came back, against what was sent (class-1 speech bits are protected, the 32 class-2 bits of a frame are not);
    # rank 0's block starts at global burst 0, so its first bursts are the distinct ones in their order
    m_sp, m_fg = min(n_s, wl["speech"].size), min(n_f // 4, wl["l2"].shape[0])
    h_fr = W["frames"].cpu().numpy()[:m_sp]
    c1 = (h_fr[:, :, :6] == wl["frames"][:m_sp, :, :6]).all(axis=(1, 2))
    h_l2, h_crc = W["l2f"].cpu().numpy()[:m_fg], W["crc_f"].cpu().numpy()[:m_fg]
    good = h_crc == 0
    out["checks"].update(speech_class1_recovered_frac=float(c1.mean()), facch3_crc_pass_frac=float(good.mean()),
                         facch3_payloads_match_sent=bool(np.array_equal(h_l2[good], wl["l2"][:m_fg][good])))
    # ~59 % at every SNR is not a decoder defect: it is the reference's sync search, reproduced on purpose
    sent1 = wl["sync_id"][:m_fg] == 1
    out["checks"].update(
        facch3_crc_pass_frac_groups_sent_with_sequence_0=float(good[~sent1].mean()) if (~sent1).any() else None,
        facch3_crc_pass_frac_groups_sent_with_sequence_1=float(good[sent1].mean()) if sent1.any() else None,
        facch3_pass_explained="the reference never clears its sync-correlation accumulator between training sequences "
                              "(pi4cxpsk.c:206-237): the NT3 FACCH format's second sequence is ranked on |c0| + |c1| and always "
                              "wins, so the groups SENT with sequence 0 (half of this workload) are demodulated against the "
                              "wrong phase reference and mostly fail; oracle and product reproduce that bit for bit.  With the "
                              "sent sequence forced both decode ~100 % at 20 dB: "
                              "tests/test_gpu_l1_tch.py::test_facch3_pass_rate_is_the_reference_sync_ranking_not_the_decoder")
    if not args.no_cpu and world == 1:
        import oracle_lib
        oracle_lib.lib()
        m_s, m_g = min(90_000, m_sp), min(2_500, m_fg)   # the first 100 000 bursts: about 10 s
        w_iq = wl["iq"].reshape(-1, stride)
        tc = time.perf_counter()
        ref_fr = np.zeros((m_s, 2, 10), np.uint8)
        # a burst's soft bits are within 1 LSB of the oracle's, except where a symbol's phase lies within the 1e-4 soft-symbol
        # tolerance of the midpoint between two constellation points: there the weakest possible soft bit (|value| = 63)
        # comes out with the other sign.  Such bursts are counted, and it is checked that this is all that differs.
        n_cmp = n_off = n_pick = 0

        def classify(ref_eb, got_eb, n_off, n_mid):
            dlt = np.abs(ref_eb.astype(np.int32) - got_eb.astype(np.int32))
            if dlt.max() <= 1:
                return n_off, n_mid
            bad = dlt > 1
            mid = bool((np.abs(np.abs(ref_eb[bad].astype(np.int32)) - 63) <= 1).all() and
                       (np.abs(np.abs(got_eb[bad].astype(np.int32)) - 63) <= 1).all())
            return n_off + 1, n_mid + int(mid)

        h_eb = W["eb_s"].cpu().numpy()
        for k in range(m_s):
            i = wl["speech"][k]
            r = oracle_lib.demod("nt3_speech", w_iq[i, :474], 4, float(wl["freq_shift"][i]))
            n_cmp += 1
            n_off, n_pick = classify(r["ebits"], h_eb[k], n_off, n_pick)
            f0, f1, _, _, _ = oracle_lib.tch3_decode(h_eb[k][None], 0)
            ref_fr[k, 0], ref_fr[k, 1] = f0[0], f1[0]
        ref_l2 = np.zeros((m_g, 10), np.uint8)
        ref_crc = np.zeros(m_g, np.int32)
        h_ebf = W["eb_f"].cpu().numpy()
        for g in range(m_g):
            for j in range(4):
                i = wl["facch"][4 * g + j]
                r = oracle_lib.demod("nt3_facch", w_iq[i, :474], 4, float(wl["freq_shift"][i]))
                n_cmp += 1
                n_off, n_pick = classify(r["ebits"], h_ebf[4 * g + j], n_off, n_pick)
            o = oracle_lib.facch3_decode(h_ebf[4 * g:4 * g + 4][None])
            ref_l2[g], ref_crc[g] = o[0][0], o[2][0]
        tc = time.perf_counter() - tc
        m = m_s + 4 * m_g
        out["cpu_baseline"] = {"value": m / tc / 1e6, "unit": "Mbursts/s", "cores": 1, "kind": "port",
                               "sample": f"first {m} bursts (demod + layer 1), gcc -O2 oracle via ctypes (per-burst calls), "
                                         f"1 thread, {tc:.1f} s"}
        out["checks"].update(soft_bits_within_1_of_oracle_frac=1.0 - n_off / n_cmp, bursts_compared=n_cmp,
                             bursts_with_other_soft_bits=n_off, of_which_only_midpoint_symbols=n_pick,
                             speech_frames_identical_to_oracle=bool(np.array_equal(h_fr[:m_s], ref_fr)),
                             facch3_identical_to_oracle=bool(np.array_equal(h_crc[:m_g], ref_crc) and
                                                             np.array_equal(h_l2[:m_g][ref_crc == 0], ref_l2[ref_crc == 0])))
    emit(out, flush=True)
    if grouped:
        import torch.distributed as dist
        dist.destroy_process_group()


def time_legacy_calls(api, wl, oracle_lib, m=600):
    """The path an UNCHANGED gmr1_rx.c takes: per burst one blocking gmr1_pi4cxpsk_demod and one gmr1_bcch_decode /
    gmr1_ccch_decode (reference rx_bcch / rx_ccch, gmr1_rx.c:746-850), host pointers in, host pointers out -- timed over
    the first `m` bursts of the workload through ctypes (its ~2 us per call are in the figure), next to the CPU oracle
    making the same two calls.  tools/legacy_loop.c measures the same from a C program."""
    import ctypes as C
    L = api.load()
    f_demod, f_dec = L.gmr1_pi4cxpsk_demod, (L.gmr1_bcch_decode, L.gmr1_ccch_decode)
    for f in (f_demod,) + f_dec:
        f.restype = C.c_int
    bt = [C.addressof(C.c_void_p.in_dll(L, "gmr1_bcch_burst")), C.addressof(C.c_void_p.in_dll(L, "gmr1_dc6_burst"))]
    eb = (C.c_int8 * 432)()
    l2 = np.zeros((m, 24), np.uint8)
    crc = np.zeros(m, np.int32)
    sid, conv, toa, fe = C.c_int(), C.c_int(), C.c_float(), C.c_float()
    vecs = []
    for i in range(m):
        k = int(wl["kind"][i])
        ln = 976 if k else 1016
        o = int(wl["offset"][i])
        vecs.append((k, api.CxVec(ln, ln, 0, wl["iq"][o:o + ln].ctypes.data_as(C.c_void_p)), l2[i].ctypes.data_as(C.c_void_p)))

    def loop():
        for i, (k, v, pl2) in enumerate(vecs):
            rc = f_demod(C.c_void_p(bt[k]), C.byref(v), C.c_int(4), C.c_float(0.0), eb, C.byref(sid), C.byref(toa), C.byref(fe))
            crc[i] = f_dec[k](pl2, eb, C.byref(conv)) if rc == 0 else -100
    loop()                                                  # one-time set-up (pinned block, stream)
    t0 = time.perf_counter()
    loop()
    t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref_l2 = np.zeros((m, 24), np.uint8)
    ref_crc = np.zeros(m, np.int32)
    for i, (k, v, _) in enumerate(vecs):
        o = int(wl["offset"][i])
        r = oracle_lib.demod("dc6" if k else "bcch", wl["iq"][o:o + (976 if k else 1016)], 4)
        if r["rv"] == 0:
            d = (oracle_lib.ccch_decode if k else oracle_lib.bcch_decode)(r["ebits"][None, :432 if k else 424])
            ref_l2[i], ref_crc[i] = d[0][0], d[1][0]
        else:
            ref_crc[i] = -100
    t_cpu = time.perf_counter() - t0
    ok = (crc == 0) | (ref_crc == 0)
    return {"us_per_demod_decode_pair": t_gpu / m * 1e6, "oracle_us_per_pair": t_cpu / m * 1e6, "bursts": m,
            "identical_to_oracle": bool(np.array_equal(crc, ref_crc) and np.array_equal(l2[ok], ref_l2[ok])),
            "note": "blocking one-burst calls with host pointers, through ctypes on both sides; a carrier-minute is about 1300 pairs"}


def rocm_smi_state(dev_index):
    """Clocks, power and performance level as the amdgpu driver reports them -- the sysfs files rocm-smi itself reads (no child
    process: a program that has touched the GPU must not start one that execs).  None where nothing is readable."""
    import glob
    out = {}

    def rd(path):
        try:
            with open(path) as fh:
                return fh.read().strip()
        except OSError:
            return None
    cards = sorted(c for c in glob.glob("/sys/class/drm/card[0-9]*/device") if rd(os.path.join(c, "vendor")) == "0x1002")
    if dev_index >= len(cards):
        return None
    dev = cards[dev_index]
    for key, name in (("sclk", "pp_dpm_sclk"), ("mclk", "pp_dpm_mclk")):
        txt = rd(os.path.join(dev, name))
        if txt:
            cur = [ln.split(":")[1].strip().rstrip("*").strip() for ln in txt.splitlines() if ln.rstrip().endswith("*")]
            out[key + "_current"] = cur[0] if cur else None
            out[key + "_levels"] = [ln.split(":")[1].strip().rstrip("*").strip() for ln in txt.splitlines() if ":" in ln]
    lvl = rd(os.path.join(dev, "power_dpm_force_performance_level"))
    if lvl:
        out["performance_level"] = lvl
    for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
        for key, name in (("power_average_w", "power1_average"), ("power_cap_w", "power1_cap"), ("power_input_w", "power1_input")):
            v = rd(os.path.join(hw, name))
            if v and v.isdigit():
                out[key] = int(v) / 1e6
    return out or None


SIDE_WORKLOADS = (
    # key, bench arguments, what "identical to the oracle" means in that workload's checks
    ("nt3", ["--workload", "nt3"], lambda c: bool(c.get("speech_frames_identical_to_oracle") and c.get("facch3_identical_to_oracle"))),
    ("tch3", ["--workload", "tch3"], lambda c: bool(c.get("frames_identical_to_oracle") and c.get("conv_identical"))),
    ("rx_64x60s", ["--workload", "rx"], lambda c: bool(c.get("frames_identical_to_oracle"))),
    ("fcch", ["--workload", "fcch"], lambda c: bool(c.get("toa_identical_to_oracle"))),
    ("chan", ["--workload", "chan"], lambda c: c.get("max_abs_err_vs_oracle") is not None and c["max_abs_err_vs_oracle"] < 2e-4),
)


def start_side_runner(args):
    """A helper process that will run the side workloads when told to.  It is started BEFORE this process touches the GPU
    (a process that has initialised the GPU must not exec another program, nor should its forked children) and never touches
    the GPU itself: it only starts `python bench.py --workload X` children, one at a time, after the timed region."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--side-runner", "--steps", str(args.steps), "--preroll-s", str(args.preroll_s),
           "--conv-decoder", args.conv_decoder]
    try:
        # (its own session = its own process group: a timeout ends the runner AND the `bench.py --workload X` child it is waiting
        # for, which would otherwise stay on the GPU under whatever runs next)
        return subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, start_new_session=True)
    except OSError as e:
        print(f"bench.py: no side-workload runner: {e!r}", file=sys.stderr)
        return None


SIDE_CHILD_TIMEOUT_S = 100


def collect_side(runner, timeout=None):
    if runner is None:
        return {"error": "the side-workload runner could not be started"}
    if timeout is None:
        timeout = len(SIDE_WORKLOADS) * SIDE_CHILD_TIMEOUT_S + 30      # never shorter than what the runner allows its children
    try:
        out, _ = runner.communicate("go\n", timeout=timeout)
        return json.loads(out.strip().splitlines()[-1])
    except Exception as e:
        import signal
        try:
            os.killpg(runner.pid, signal.SIGKILL)      # the runner's whole group (start_new_session): its current child too
        except Exception:
            try:
                runner.kill()
            except Exception:
                pass
        try:
            runner.wait(timeout=10)
        except Exception:
            pass
        return {"error": repr(e)}


def side_runner_main(args):
    """`bench.py --side-runner`: wait for "go" on stdin, run the side workloads, print their JSON."""
    if sys.stdin.readline().strip() != "go":
        return
    print(json.dumps(side_workloads(args)), flush=True)


def side_workloads(args):
    """The other BASELINE configs at their full sizes, each as `python bench.py --workload X` in a child process (the GPU is
    idle by now; one child at a time): {ms, frac, identical_to_oracle, ...} per workload."""
    import subprocess
    side = {}
    t_all = time.perf_counter()
    for key, argv, same in SIDE_WORKLOADS:
        # (the same untimed pre-roll and warm-up as the headline: these children start on a GPU that sat idle through the CPU
        # baseline, and with 0.2 s / 3 steps the channelizer read 8 % slower here than in its own bench line)
        cmd = [sys.executable, os.path.abspath(__file__)] + argv + ["--steps", str(min(args.steps, 50)), "--warmup", str(min(args.warmup, 10)),
                                                                    "--preroll-s", str(args.preroll_s),
                                                                    "--conv-decoder", args.conv_decoder]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=SIDE_CHILD_TIMEOUT_S)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            rf = d.get("roofline") or {}
            side[key] = {"ms": d["ms_per_step"], "value": d["value"], "unit": d["unit"], "frac": rf.get("frac"),
                         # what the workload is actually limited by (its own line's roofline.limited_by); `frac` is always the
                         # algorithmic bytes against the 8 TB/s HBM roof, nominal where the limit is something else
                         "bound": rf.get("limited_by") or rf.get("bound"), "frac_is_of": "hbm 8 TB/s",
                         "identical_to_oracle": same(d.get("checks") or {}), "workload": (d.get("config") or {}).get("workload"),
                         "wall_s": round(time.perf_counter() - t0, 1)}
            for k in ("gpu_vs_all_cores", "phases_ms"):
                if k in d:
                    side[key][k] = d[k]
            if isinstance(d.get("roofline_valu"), dict) and "valu_busy" in d["roofline_valu"]:
                side[key]["valu_busy"] = d["roofline_valu"]["valu_busy"]
                side[key]["avg_resident_waves"] = d["roofline_valu"].get("avg_resident_waves")
            if key == "chan":
                side[key]["max_abs_err_vs_oracle"] = (d.get("checks") or {}).get("max_abs_err_vs_oracle")
                side[key]["identical_to_oracle_means"] = "within 2e-4 of the numpy oracle (floating point)"
        except Exception as e:
            side[key] = {"error": repr(e), "wall_s": round(time.perf_counter() - t0, 1)}
    side["wall_s"] = round(time.perf_counter() - t_all, 1)
    side["what"] = ("every other BASELINE config on this box, after the timed region, one `python bench.py --workload X` child "
                    "each: ms per step, fraction of the 8 TB/s roof, and whether the outputs equal the CPU oracle's")
    return side


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv=None, script=None, extra_env=None, timeout=None):
    """`python bench.py --gpus N` without a launcher: the parent starts N fresh interpreters (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, the same command line) and waits for them.  The
    parent never imports torch and never makes a HIP call (a process that has touched the GPU must not fork / exec
    others on this pool); children inherit stdout, rank 0 prints the one JSON line.  Returns the worst exit code."""
    import subprocess
    argv = list(sys.argv[1:] if argv is None else argv)
    script = script or os.path.abspath(__file__)
    env = dict(os.environ)
    env.update({"WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                "MASTER_PORT": str(free_port()), "GMR1_BENCH_SPAWNED": "1"})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra_env or {})
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, script] + argv, env=e))
    t_end = None if timeout is None else time.time() + timeout
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:
                    rc = rc or code
                    # a rank that died leaves the others waiting in a collective: end exactly those children
                    for q in pending:
                        q.terminate()
            if t_end is not None and time.time() > t_end:
                rc = rc or 124
                for q in pending:
                    q.terminate()
                t_end = None
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def sharded_rx_extra(args, pkg, dev, backend, rank, world, partial):
    """The north star's config-4 exchange, measured next to the headline (N > 1 only; keys outside `value`):
    rank 0 holds the channelised capture (--shard-arfcns carriers x --shard-seconds, device-resident), carrier a goes
    to rank a mod N point to point (RCCL send / recv over xGMI under backend nccl), every rank runs the receive loop
    (gmr1_hip_rx_run_dev) on its carriers, the 40-byte frame records come back to rank 0 (SURVEY.md 8e;
    independence per gmr1_rx.c:732-741).  Returns the dict for the JSON line on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    import workloads
    sh, api = pkg.shard, pkg.api
    A, sps, seconds = args.shard_arfcns, 4, args.shard_seconds
    ns = int(seconds * 23400 * sps)
    distinct = min(A, 8)
    host, slices = None, None
    cdev = dev if backend == "nccl" else None            # gloo moves host tensors
    if rank == 0:
        host = [workloads.bcch_carrier(pkg, 700 + a, seconds=seconds, sps=sps, stn=(5 * a) % 24, delay=a % 8,
                                       cfo_hz=40.0 * (a - 3), esn0_db=10.0 + a)[0] for a in range(distinct)]
        base = [torch.from_numpy(h) for h in host]
        if cdev is not None:
            base = [b.to(cdev) for b in base]
        slices = [base[a % distinct] for a in range(A)]
    t = {}
    out = None
    for it in range(2):                                  # pass 0 warms the communicator and the kernels
        dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        mine = sh.scatter_iq(slices, A, ns, src=0, device=cdev)
        if dev.type == "cuda":
            torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        rec, key = sh.rx_run_on_slices(api, mine, ns, sps=sps, device=dev, with_key=True,
                                           max_records=max(len(mine), 1) * 4096)
        t1b = time.perf_counter()
        dist.barrier()
        t2 = time.perf_counter()
        out = sh.gather_records(rec, dst=0, device=cdev, order_key=key)
        dist.barrier()
        t3 = time.perf_counter()
        t = {"scatter_ms": (t1 - t0) * 1e3, "rx_loop_ms": (t2 - t1) * 1e3, "gather_ms": (t3 - t2) * 1e3,
             "rx_only_ms": (t1b - t1) * 1e3}
    tdev = dev if backend == "nccl" else "cpu"
    tt = torch.tensor([t["scatter_ms"], t["rx_loop_ms"], t["gather_ms"]], dtype=torch.float64, device=tdev)
    # the receive loop alone, per rank, without the barrier behind it: its spread over the ranks
    lo = torch.tensor([t["rx_only_ms"]], dtype=torch.float64, device=tdev)
    hi = lo.clone()
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)

    # ---- the other way to feed the ranks: every rank already holds the carriers it owns (a recorder hands each GPU its
    # carriers; here every rank synthesises its own).  No scatter; same loop, same gather.
    own_ids = sh.my_arfcns(A, world, rank)
    need = sorted({a % distinct for a in own_ids})
    gen = {d: torch.from_numpy(workloads.bcch_carrier(pkg, 700 + d, seconds=seconds, sps=sps, stn=(5 * d) % 24, delay=d % 8,
                                                      cfo_hz=40.0 * (d - 3), esn0_db=10.0 + d)[0]) for d in need}
    if cdev is not None:
        gen = {d: g.to(cdev) for d, g in gen.items()}
    resident = {a: gen[a % distinct] for a in own_ids}
    tr = {}
    out_res = None
    for it in range(2):
        dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        rec, key = sh.rx_run_on_slices(api, resident, ns, sps=sps, device=dev, with_key=True, max_records=max(len(resident), 1) * 4096)
        t1 = time.perf_counter()
        dist.barrier()
        t2 = time.perf_counter()
        out_res = sh.gather_records(rec, dst=0, device=cdev, order_key=key)
        dist.barrier()
        t3 = time.perf_counter()
        tr = {"rx_loop_ms": (t2 - t0) * 1e3, "gather_ms": (t3 - t2) * 1e3, "rx_only_ms": (t1 - t0) * 1e3}
    tr_t = torch.tensor([tr["rx_loop_ms"], tr["gather_ms"]], dtype=torch.float64, device=tdev)
    rlo = torch.tensor([tr["rx_only_ms"]], dtype=torch.float64, device=tdev)
    rhi = rlo.clone()
    dist.all_reduce(tr_t, op=dist.ReduceOp.MAX)
    dist.all_reduce(rlo, op=dist.ReduceOp.MIN)
    dist.all_reduce(rhi, op=dist.ReduceOp.MAX)

    res = None
    if rank == 0:
        res = _sharded_summary(args, dist, host, out, tt, A, ns, seconds, distinct, world, sps)
        res["rx_loop_ms_per_rank_min_max"] = [float(lo[0]), float(hi[0])]
        key = lambda r: [(int(x["arfcn"]), int(x["chain"]), int(x["type"]), int(x["fn"]), int(x["tn"]), bytes(x["l2"])) for x in r]
        res["resident"] = {"what": "every rank holds the carriers it owns already (no scatter): receive loop + record gather",
                           "rx_loop_ms": float(tr_t[0]), "gather_ms": float(tr_t[1]),
                           "rx_loop_ms_per_rank_min_max": [float(rlo[0]), float(rhi[0])],
                           "records_identical_to_scattered_run": bool(key(out_res) == key(out))}
        # and the whole exchange against ONE gmr1_hip_rx_run over all carriers on this GPU: every record of every carrier
        if dev.type == "cuda":
            iq_all = torch.cat([torch.view_as_real(b.to(dev)).reshape(-1) for b in base]).contiguous()
            off1 = (np.arange(A, dtype=np.uint64) % np.uint64(distinct)) * np.uint64(ns)
            one, _, _, _ = api.rx_run_dev(torch.cuda.current_stream(dev).cuda_stream, iq_all.data_ptr(), off1, np.full(A, ns, np.uint64),
                                          sps=sps, arfcn=np.arange(A, dtype=np.uint16), max_records=A * 4096)
            res["records_identical_to_single_gpu_run"] = bool(key(one) == key(out))
            res["single_gpu_run_frames"] = int(len(one))
            del iq_all
        partial["sharded_rx"] = res                  # what the watchdog prints if the native leg below never returns
    if backend == "nccl":
        nat = _sharded_native(pkg, dist, dev, rank, world, A, ns, distinct, sps, base if rank == 0 else None, out,
                              resident=(resident, own_ids))
        if rank == 0:
            res["native"] = nat
    return res


def _sharded_native(pkg, dist, dev, rank, world, A, ns, distinct, sps, base, torch_records, resident=None):
    """The same exchange through the library's own C entry point (gmr1_hip_rx_run_sharded, include/gmr1_hip_shard.h):
    ncclSend / ncclRecv posted from C++, no Python between scatter, receive loop and gather."""
    import torch
    api = pkg.api
    ids = [api.Shard.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    with stdout_to_stderr():                       # a communicator of the library's own: its banner goes to stderr too
        sh = api.Shard(ids[0], rank, world)
    try:
        iq_all = torch.cat([torch.view_as_real(b).reshape(-1) for b in base]).contiguous() if rank == 0 else None
        offset = (np.arange(A, dtype=np.uint64) % np.uint64(distinct)) * np.uint64(ns)      # tiles share their samples
        length = np.full(A, ns, np.uint64)
        st = torch.cuda.current_stream(dev).cuda_stream
        rec = tim = None
        for _ in range(2):                               # the first pass warms the communicator
            dist.barrier()
            rec, status, chains, tim = sh.rx_run(st, iq_all.data_ptr() if rank == 0 else 0, offset, length, sps=sps,
                                                 arfcn=np.arange(A, dtype=np.uint16), max_records=A * 4096)
        tt = torch.tensor(tim.astype(np.float64), device=dev)
        lo, hi = tt[1:2].clone(), tt[1:2].clone()
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        # every rank's carriers already in its own memory (gmr1_hip_rx_run_sharded_resident): no scatter
        rec2 = tim2 = None
        if resident is not None:
            mine, own_ids = resident
            pos = {a: k for k, a in enumerate(own_ids)}
            mem = (torch.cat([torch.view_as_real(mine[a].to(dev)).reshape(-1) for a in own_ids]).contiguous()
                   if own_ids else torch.zeros(2, device=dev))
            off2 = np.array([pos.get(a, 0) * ns for a in range(A)], np.uint64)
            for _ in range(2):
                dist.barrier()
                rec2, _, _, tim2 = sh.rx_run(st, mem.data_ptr(), off2, length, sps=sps, arfcn=np.arange(A, dtype=np.uint16),
                                             max_records=A * 4096, resident=True)
            t2 = torch.tensor(tim2.astype(np.float64), device=dev)
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        if rank != 0:
            return None
        key = lambda r: [(int(x["arfcn"]), int(x["chain"]), int(x["type"]), int(x["fn"]), int(x["tn"]), bytes(x["l2"])) for x in r]
        res = {"entry_point": "gmr1_hip_rx_run_sharded (RCCL from C++)", "scatter_ms": float(tt[0]), "rx_loop_ms": float(tt[1]),
               "gather_ms": float(tt[2]), "rx_loop_ms_per_rank_min_max": [float(lo[0]), float(hi[0])], "frames": int(rec.size),
               "records_identical_to_torch_distributed_path": bool(key(rec) == key(torch_records))}
        if rec2 is not None:
            res["resident"] = {"entry_point": "gmr1_hip_rx_run_sharded_resident", "rx_loop_ms": float(t2[1]), "gather_ms": float(t2[2]),
                               "records_identical_to_scattered_run": bool(key(rec2) == key(rec))}
        return res
    finally:
        sh.close()


def _sharded_summary(args, dist, host, out, tt, A, ns, seconds, distinct, world, sps):
    res = {"workload": f"configs[3] sharded: {A} carriers x {seconds:g} s @ 93.6 ksps ({distinct} distinct, tiled) held by "
                       f"rank 0, carrier a -> rank a mod {world}, p2p scatter, gmr1_rx loop per rank, 40-byte records gathered",
           "backend": dist.get_backend(), "ranks_seen": dist.get_world_size(),
           "scatter_ms": float(tt[0]), "rx_loop_ms": float(tt[1]), "gather_ms": float(tt[2]),
           "scatter_bytes": int(sum(1 for a in range(A) if a % world != 0) * ns * 8),
           "frames": int(out.size), "carriers_with_frames": int(np.unique(out["arfcn"]).size)}
    res["scatter_GBps"] = res["scatter_bytes"] / max(res["scatter_ms"], 1e-9) / 1e6
    res["samples_per_s"] = A * ns / ((res["scatter_ms"] + res["rx_loop_ms"] + res["gather_ms"]) * 1e-3)
    if not args.no_cpu:
        import oracle_lib                                  # checker only: carrier 0's frames against the CPU loop
        oracle_lib.lib()
        _, orec, _ = oracle_lib.rx_run(host[0], sps=sps, arfcn=0)
        mine0 = out[out["arfcn"] == 0]
        key = lambda r: [(int(x["chain"]), int(x["type"]), int(x["fn"]), int(x["tn"]), bytes(x["l2"])) for x in r]
        res["frames_identical_to_oracle"] = bool(key(mine0) == key(orec))
        res["oracle_frames_carrier0"] = int(len(orec))
        # every tile of carrier 0 (a, a + distinct, ...) was decoded on a different rank: they must agree with it
        same = all(key(out[out["arfcn"] == a]) == key(mine0) for a in range(0, A, distinct))
        res["tiles_of_carrier0_identical_across_ranks"] = bool(same)
    return res


class stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when a communicator comes up; the contract is ONE JSON line there.  Inside
    this block file descriptor 1 points at stderr (native writes included)."""
    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def init_ranks():
    """One process per GPU: rank / world from the launcher's (or spawn_ranks') environment, the device of this rank, and
    the process group when there is more than one rank.  Returns (rank, world, backend, device, device index, grouped)."""
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE") or "1")
    # GMR1_BENCH_BACKEND=gloo lets the N > 1 code path be exercised on a box with fewer GPUs than ranks
    # (ranks then share devices and the exchanges run over gloo on host tensors); the driver never sets it
    backend = os.environ.get("GMR1_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()                      # does not initialise the GPU
    if n_dev == 0:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if backend == "nccl" and world > n_dev:
        raise SystemExit(f"bench.py: --gpus {world} but this node shows {n_dev} GPU(s); RCCL needs one device per rank "
                         "(GMR1_BENCH_BACKEND=gloo rehearses the rank logic on fewer devices)")
    dev_index = local_rank if backend == "nccl" else local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # GMR1_BENCH_FORCE_GROUP=1: a single rank goes through everything the N > 1 ranks go through (RCCL group, barriers,
    # max-over-ranks, the sharded exchange with itself) - the only way to touch RCCL on a one-GPU box
    grouped = world > 1 or os.environ.get("GMR1_BENCH_FORCE_GROUP") == "1"
    if grouped:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        with stdout_to_stderr():
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            dist.barrier()                # the communicator (and its banner) comes up with the first collective
    return rank, world, backend, dev, dev_index, grouped


def gather_bytes(arr, rank, world, backend, dev):
    """Concatenation over the ranks (in rank order) of a uint8 array of any per-rank length, on rank 0 (None elsewhere)."""
    import torch
    import torch.distributed as dist
    flat = np.ascontiguousarray(arr).reshape(-1).view(np.uint8)
    tdev = dev if backend == "nccl" else "cpu"
    sizes = [torch.zeros(1, dtype=torch.int64, device=tdev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([flat.size], dtype=torch.int64, device=tdev))
    sizes = [int(x.item()) for x in sizes]
    m = max(sizes + [1])
    buf = torch.zeros(m, dtype=torch.uint8, device=tdev)
    buf[:flat.size] = torch.from_numpy(flat.copy()).to(tdev)
    parts = [torch.zeros(m, dtype=torch.uint8, device=tdev) for _ in range(world)]
    dist.all_gather(parts, buf)
    if rank != 0:
        return None
    return np.concatenate([p[:k].cpu().numpy() for p, k in zip(parts, sizes)])


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if args.workload not in ("bursts", "nt3", "tch3") and (args.gpus > 1 or int(env_world or "1") > 1):
        raise SystemExit(f"bench.py: --workload {args.workload} is a one-GPU side measurement; N-GPU lines exist for the "
                         "headline workload (drop --workload) and for configs[4] (--workload nt3 | tch3)")
    if env_world is None and args.gpus > 1:
        # no launcher: start the N ranks ourselves, before anything in this process touches the GPU
        raise SystemExit(spawn_ranks(args.gpus))
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={env_world} but --gpus {args.gpus}: refusing to report a line for the "
                         "wrong number of GPUs (start it as `python bench.py --gpus N`, or with a launcher whose world "
                         "size equals N)")
    if args.side_runner:
        return side_runner_main(args)
    if args.workload == "nt3":
        return run_nt3_workload(args)
    if args.workload == "rx":
        return run_rx_workload(args)
    if args.workload == "chan":
        return run_chan_workload(args)
    if args.workload == "ambe":
        return run_ambe_workload(args)
    if args.workload != "bursts":
        return run_side_workload(args)
    # every other BASELINE config goes into the same record (`side`): their runner starts now, before the GPU is touched
    side_runner = None
    # (not under a profiler: its preloaded library initialises the GPU in every process before main(), and a process that has
    # done so must not start children that exec)
    profiled = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if (env_world is None and args.gpus == 1 and not args.no_extras and not args.no_cpu and not profiled
            and not os.environ.get("GMR1_BENCH_FORCE_GROUP")):
        side_runner = start_side_runner(args)
    import torch
    from __graft_entry__ import load_package
    import workloads

    rank, world, backend, dev, dev_index, grouped = init_ranks()

    pkg = load_package()
    api = pkg.api
    api.load()
    api.init(dev_index)

    n = args.bursts
    t_gen = time.time()
    wl = workloads.bcch_ccch_mix(pkg, n=n, seed=3 + rank)
    t_gen = time.time() - t_gen

    # ---- inputs to HBM (outside the timed region) ---------------------------------------
    iq = torch.from_numpy(wl["iq"].view(np.float32)).to(dev)
    offset = torch.from_numpy(wl["offset"].astype(np.int64)).to(dev)
    kind = torch.from_numpy(wl["kind"]).to(dev)
    l2 = torch.zeros((n, 24), dtype=torch.uint8, device=dev)
    crc = torch.zeros(n, dtype=torch.int32, device=dev)
    conv = torch.zeros(n, dtype=torch.int32, device=dev)
    toa = torch.zeros(n, dtype=torch.float32, device=dev)
    ferr = torch.zeros(n, dtype=torch.float32, device=dev)
    rv = torch.zeros(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)

    # the same samples polyphase-planar (sample s at planes[(s & 3) * P + (s >> 2)]): what the opt-in entry point reads
    n_samp = wl["iq"].size
    P = -(-n_samp // 4)
    want_planar = args.layout == "planar" or not args.no_extras
    planes = None
    if want_planar:
        planes = torch.zeros(4 * P * 2, dtype=torch.float32, device=dev)
        api.iq_to_planar_dev(stream.cuda_stream, 4, n_samp, iq.data_ptr(), planes.data_ptr(), P)

    def step_interleaved():
        api.rx_bcch_ccch_batch_dev(stream.cuda_stream, n, 4, iq.data_ptr(), offset.data_ptr(), kind.data_ptr(),
                                   None, l2.data_ptr(), crc.data_ptr(), conv.data_ptr(), toa.data_ptr(),
                                   ferr.data_ptr(), None, None, rv.data_ptr())

    def step_planar():
        api.rx_bcch_ccch_batch_planar_dev(stream.cuda_stream, n, 4, planes.data_ptr(), P, offset.data_ptr(), kind.data_ptr(),
                                          None, l2.data_ptr(), crc.data_ptr(), conv.data_ptr(), toa.data_ptr(),
                                          ferr.data_ptr(), None, None, rv.data_ptr())
    step = step_planar if args.layout == "planar" else step_interleaved

    def time_steps(fn, k):
        """K launches of `fn` between two events on the launch stream -> ms per launch (after the timed region only)."""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # the same untimed pre-roll as the headline's timed region: the result checks before this left the GPU idle, and
        # a measurement that starts at idle clocks reads about 4 % slow
        preroll(fn, args.preroll_s)
        for _ in range(max(3, args.warmup)):
            fn()
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(k):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / k

    def barrier():
        if grouped:
            import torch.distributed as dist
            dist.barrier()

    preroll(step, args.preroll_s)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    clock = {}
    try:
        clock["shader_mhz_before_timed_region"] = round(api.clock_probe_dev(stream.cuda_stream, 100)[0], 1)
        # the probe is one wave spinning for 0.1 ms on an otherwise idle GPU: three steps behind it were NOT enough to be back in
        # the steady state -- whatever was timed right behind them read 2 % slower than the same kernel timed later in the run
        # (either Viterbi decoder, whichever came first: r06t / r06x `other_decoder`).  A short pre-roll again, untimed like the
        # first (declared in `config`), then the W warm-up steps once more.
        preroll(step, min(args.preroll_s, 0.1))
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
    except Exception as e:
        clock["error"] = repr(e)

    # ---- timed region ---------------------------------------------------------------------
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)            # same stream the kernels are launched on
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    wall = t1 - t0
    kern_ms = ev0.elapsed_time(ev1) / args.steps      # one launch per step
    # the shader clock the steps just ran at, read on the device right behind them (DVFS moves in milliseconds), and what
    # the driver says about the part: boxes of the pool hold different clocks under this load
    try:
        mhz, wall_mhz = api.clock_probe_dev(stream.cuda_stream, 100)
        clock["shader_mhz_after_timed_region"] = round(mhz, 1)
        clock["wall_counter_mhz"] = wall_mhz
        props = torch.cuda.get_device_properties(dev)
        clock["device"] = props.name
        clock["max_shader_mhz"] = getattr(props, "clock_rate", 0) / 1e3 or None
        clock["compute_units"] = props.multi_processor_count
    except Exception as e:
        clock["error"] = repr(e)

    if grouped:
        import torch.distributed as dist
        tt = torch.tensor([wall], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())

    # profiling build only: how many bursts of one launch left the speculated-pick path of k_rx4
    missed = None
    if os.environ.get("GMR1_HIP_DBG_STOP") == "105":      # (the count costs time: only when asked for)
        try:
            f_miss = api.load().gmr1_hip_prof_miss
            f_miss()
            step()
            torch.cuda.synchronize()
            missed = int(f_miss())
        except AttributeError:
            pass

    # ---- sanity on the produced results (outside the timed region) --------------------------
    h_crc = crc.cpu().numpy()
    h_l2 = l2.cpu().numpy()
    good = h_crc == 0
    payload_ok = bool(np.array_equal(h_l2[good], wl["l2"][good]))
    decoded_frac = float(good.mean())

    # ---- after the timed region, outside `value`: the opt-in planar layout and the other Viterbi decoder ----------
    extras = {}
    if not args.no_extras and args.layout == "interleaved":
        h_all = [t.cpu().numpy().copy() for t in (crc, conv, toa, ferr, rv)]
        ms_pl = time_steps(step_planar, args.steps)
        same = bool(np.array_equal(l2.cpu().numpy(), h_l2) and
                    all(np.array_equal(t.cpu().numpy().view(np.uint8), h.view(np.uint8)) for t, h in zip((crc, conv, toa, ferr, rv), h_all)))
        extras["planar"] = (ms_pl, same)
        cur = api.get_conv_decoder()
        other = api.CONV_GENERIC if cur == api.CONV_ACC else api.CONV_ACC
        with api.conv_decoder(other):
            ms_other = time_steps(step_interleaved, args.steps)
            ms_other_pl = time_steps(step_planar, args.steps)
        extras["other_decoder"] = ("acc" if other == api.CONV_ACC else "generic", ms_other, ms_other_pl)
        step_interleaved()                       # leave the default decoder's results in the buffers
        torch.cuda.synchronize()

    # ---- N > 1: the north star's exchange (scatter of IQ slices, receive loop, gather of frames) ----
    # Extra keys only, after the timed region.  A watchdog prints the headline line without them if the exchange
    # does not come back (a collective that hangs must not cost the measurement that is already taken).
    sharded = None
    line = {}
    if grouped and not args.no_shard:
        import threading

        def give_up():
            if rank == 0 and line:
                part = line.get("sharded_rx") or {}
                part["error"] = f"not finished after {args.shard_timeout:g} s"
                line["sharded_rx"] = part
                emit(line, flush=True)
            # the headline part of the line is valid and printed, but an exchange that hung is a failed run: every rank
            # says so with its exit code (spawn_ranks and a launcher then report it), not only with the "error" key
            os._exit(3)
        dog = threading.Timer(args.shard_timeout, give_up)
        dog.daemon = True
    else:
        dog = None

    def finish():
        if grouped:
            import torch.distributed as dist
            dist.destroy_process_group()

    if rank != 0:
        if dog is not None:
            dog.start()
            try:
                sharded_rx_extra(args, pkg, dev, backend, rank, world, partial={})
            except Exception as e:                      # rank 0 reports; this rank has nothing to print
                print(f"rank {rank}: sharded config-4 run failed: {e!r}", file=sys.stderr)
            dog.cancel()
        finish()
        return

    n_bcch = int((wl["kind"] == 0).sum())
    bytes_per_launch = n_bcch * BYTES_BCCH + (n - n_bcch) * BYTES_CCCH
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
    traffic, traffic_planar, traffic_note = None, None, "profiles/hbm_traffic.json missing"
    tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tf):
        try:
            tj = json.load(open(tf))
            if tj.get("kernel_sources_sha256") != kernel_sources_hash():
                traffic_note = "profiles/hbm_traffic.json was taken on other kernel sources: not reported"
            elif n != 100_000:
                traffic_note = "profiles/hbm_traffic.json is per 100000 bursts"
            else:
                traffic = tj.get("k_rx_planar_bytes_per_launch_100k" if args.layout == "planar" else "k_rx_bytes_per_launch_100k")
                traffic_planar = tj.get("k_rx_planar_bytes_per_launch_100k")
                traffic_note = (f"builder-measured, hash-gated file profiles/hbm_traffic.json (not measured in this run): PMC FETCH_SIZE x2 + "
                                f"WRITE_SIZE, separate passes, build {tj.get('tag')}")
        except Exception as e:
            traffic_note = f"profiles/hbm_traffic.json unreadable: {e!r}"

    out = {
        "metric": "Mbursts/s demod+Viterbi (and IQ Msamp/s), 1/2/4/8 MI355X",
        "value": world * n * args.steps / wall / 1e6,
        "unit": "Mbursts/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32+i32", "data": "synthetic",
        "config": {"workload": "configs[2]: batch of 100k normal bursts, pi4cxpsk demod + rate-1/2 K=5 "
                               "Viterbi (BCCH:CCCH 1:6), sps=4",
                   "bursts_per_gpu": n, "global_bursts": world * n, "sps": 4, "untimed_preroll_s": args.preroll_s,
                   "untimed_preroll_after_clock_probe_s": min(args.preroll_s, 0.1),
                   "parallelism": f"bursts sharded over {world} rank(s), no collective"},
        "iq_msamp_per_s": world * (n_bcch * 1016 + (n - n_bcch) * 976) * args.steps / wall / 1e6,
        "roofline": {"bound": "hbm", "kernel": "k_rx4<16,4>" + (" (planar)" if args.layout == "planar" else ""), "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": traffic_note, "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": bytes_per_launch},
        "checks": {"crc_pass_frac": decoded_frac, "payloads_match_sent": payload_ok,
                   "workload_gen_s": round(t_gen, 1),
                   **({"mis_speculated_picks_per_launch": missed} if missed is not None else {})},
        "clock": clock,
    }
    smi = rocm_smi_state(dev_index)
    if smi:
        out["clock"]["driver"] = smi
    out["config"]["layout"] = args.layout
    # What the HBM fraction above is up against: the kernel's vector ALUs, machine-wide (PMC summary of the same kernel on the
    # same sources, tools/pmc_rx4.sh -> tools/valu_summary.py; like the traffic figure it is reported only while the sources'
    # hash matches)
    if args.layout == "interleaved":
        vp = valu_profile("k_rx4")
        if vp and "valu_busy" in vp:
            out["roofline_valu"] = dict({"bound": "valu_issue",
                                         "hbm_frac_if_valu_were_100pct_busy": achieved / HBM_PEAK_GBS / vp["valu_busy"],
                                         "note": "share of the launch a SIMD's vector ALU is executing, and waves resident per SIMD, both "
                                                 "averaged over the whole launch (ramp and drain included); at this instruction count "
                                                 "the HBM fraction cannot pass the figure above"}, **vp)
        elif vp:
            out["roofline_valu"] = vp
    if "planar" in extras:
        ms_pl, same = extras["planar"]
        ach = bytes_per_launch / (ms_pl * 1e-3) / 1e9
        out["roofline_planar"] = {"what": "the same step through gmr1_hip_rx_bcch_ccch_batch_planar_dev on the same samples stored "
                                          "polyphase-planar (opt-in layout; the conversion is outside the step, as the channelizer "
                                          "can write the layout directly); never part of `value`",
                                  "bound": "hbm", "kernel": "k_rx4<16,4> (planar)", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": ach / HBM_PEAK_GBS, "traffic": traffic_planar, "kernel_ms": ms_pl,
                                  "algorithmic_bytes_per_launch": bytes_per_launch,
                                  "outputs_bit_identical_to_interleaved": same}
    if "other_decoder" in extras:
        name, ms_o, ms_o_pl = extras["other_decoder"]
        out["other_decoder"] = {"conv_decoder": name, "kernel_ms": ms_o, "frac": bytes_per_launch / (ms_o * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "planar_kernel_ms": ms_o_pl, "planar_frac": bytes_per_launch / (ms_o_pl * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "what": "the same step with gmr1_hip_set_conv_decoder set to the other libosmocore decoder, timed "
                                        "after the timed region on the same box (boxes differ by a few per cent, the two modes on "
                                        "one box do not)"}

    # ---- CPU baseline: the oracle (a port, 1 core) on a bounded sample ----------------------
    if world == 1 and not args.no_cpu:
        import oracle_lib
        m = min(n, args.cpu_sample)
        end = int(wl["offset"][m]) if m < n else wl["iq"].size
        oracle_lib.lib()
        passes = max(1, args.cpu_passes)
        tc = time.perf_counter()
        for _ in range(passes):
            ref = oracle_lib.demod_decode_batch(wl["iq"][:end], wl["offset"][:m], wl["kind"][:m], sps=4,
                                                want_ebits=False, want_ssyms=False)
        tc = time.perf_counter() - tc
        same_crc = bool(np.array_equal(ref["crc"], h_crc[:m]))
        ok = (ref["crc"] == 0) | (h_crc[:m] == 0)
        same_l2 = bool(np.array_equal(ref["l2"][ok], h_l2[:m][ok]))
        out["cpu_baseline"] = {"value": passes * m / tc / 1e6, "unit": "Mbursts/s", "cores": 1, "kind": "port",
                               "sample": f"{passes} passes over the first {m} bursts of the same workload, gcc -O2 oracle, "
                                         f"1 thread, {tc:.1f} s"}
        # the same port on all host cores (the reference itself is single-threaded: one process per capture): threads over
        # contiguous burst ranges, each calling the C oracle (ctypes releases the GIL for the call)
        from concurrent.futures import ThreadPoolExecutor
        cores = max(1, min(len(os.sched_getaffinity(0)), 16))       # a one-GPU box's CPU share is 16 cores
        reps = 12                                    # every thread walks its slice `reps` times: a few seconds in total
        bounds = np.linspace(0, m, cores + 1).astype(int)

        def work(t):
            lo, hi = int(bounds[t]), int(bounds[t + 1])
            if hi <= lo:
                return
            e = int(wl["offset"][hi]) if hi < n else wl["iq"].size
            for _ in range(reps):
                oracle_lib.demod_decode_batch(wl["iq"][:e], wl["offset"][lo:hi], wl["kind"][lo:hi], sps=4,
                                              want_ebits=False, want_ssyms=False)
        ta = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(work, range(cores)))
        ta = time.perf_counter() - ta
        out["cpu_baseline_all_cores"] = {"value": reps * m / ta / 1e6, "unit": "Mbursts/s", "cores": cores, "kind": "port",
                                         "sample": f"{reps} passes over the same {m} bursts split over {cores} threads, {ta:.1f} s"}
        out["checks"]["gpu_vs_oracle_crc_identical"] = same_crc
        out["checks"]["gpu_vs_oracle_payloads_identical"] = same_l2
        out["legacy_one_burst_calls"] = time_legacy_calls(api, wl, oracle_lib)
    # ---- every other BASELINE config in the same record (never part of `value`): one subprocess each, after everything
    # above; what each reports is its own bench line's ms per step, roofline fraction and oracle comparison
    if side_runner is not None:
        out["side"] = collect_side(side_runner)
    if dog is not None:
        line.update(out)
        dog.start()
        try:
            out["sharded_rx"] = sharded_rx_extra(args, pkg, dev, backend, rank, world, partial=line)
        except Exception as e:
            out["sharded_rx"] = dict(line.get("sharded_rx") or {}, error=repr(e))
        dog.cancel()
    emit(out, flush=True)
    finish()


if __name__ == "__main__":
    main()
