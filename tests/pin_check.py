"""Checks a third_party_pins.json written by tools/pin_3p.c (the real libosmocore / libosmo-dsp on an integrator's
machine -- or, in tests/test_pin_kit.py, the oracle standing in for them) against the oracle's restatements.

For every convolutional code the file's outputs must be the generic decoder's (decision D1 / D4, oracle/orc_3p.c) or
the accelerated decoder's (D1b, oracle/orc_3p_acc.c) on ALL its vectors; the codes libosmocore can hand to its accelerated
decoder must agree on one of the two.  The DSP functions must match within float rounding, the early / late peak position
exactly (D3), normalisation as D2 states.  Returns a report; raises AssertionError with the first difference otherwise."""
import ctypes as C
import json

import numpy as np


class ConvCode(C.Structure):
    """struct orc_conv_code (oracle/orc_3p.h)"""
    _fields_ = [("N", C.c_int), ("K", C.c_int), ("len", C.c_int), ("term", C.c_int),
                ("next_output", (C.c_uint8 * 2) * 256), ("next_state", (C.c_uint8 * 2) * 256),
                ("n_punct", C.c_int), ("punct", C.c_int * 1024)]


class Cf(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


def _cv(rows):
    a = np.array(rows, np.float64)
    return np.ascontiguousarray(a[:, 0] + 1j * a[:, 1], np.complex64) if a.size else np.zeros(0, np.complex64)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def load(path):
    with open(path) as f:
        return json.load(f)


def check_conv(pins, orc):
    lib = orc.lib()
    report = {}
    for code in pins["conv"]:
        cc = ConvCode()
        polys = (C.c_uint * len(code["polys"]))(*code["polys"])
        lib.orc_conv_make(C.byref(cc), C.c_int(code["N"]), C.c_int(code["K"]), C.c_int(code["len"]), C.c_int(code["term"]), polys)
        cc.n_punct = len(code["punct"])
        for i, p in enumerate(code["punct"]):
            cc.punct[i] = p
        cc.punct[cc.n_punct] = -1
        lib.orc_conv_output_length.restype = C.c_int
        assert lib.orc_conv_output_length(C.byref(cc)) == code["n_in"], (code["name"], "coded length")
        acc_ok = code["K"] in (5, 7) and 2 <= code["N"] <= 4
        match = {"generic": 0, "acc": 0}
        first_bad = None
        for vi, vec in enumerate(code["vectors"]):
            sym = np.frombuffer(bytes.fromhex(vec["in"]), np.int8).copy()
            assert sym.size == code["n_in"]
            out = np.zeros(code["len"], np.uint8)
            with orc.conv_mode(0):
                rv0 = lib.orc_conv_decode(C.byref(cc), _ptr(sym), _ptr(out))
            out0 = "".join(map(str, out))
            is0 = out0 == vec["out"] and rv0 == vec["rv"]
            is1 = False
            if acc_ok:
                out[:] = 0
                rv1 = lib.orc_conv_decode_acc(C.byref(cc), _ptr(sym), _ptr(out))
                is1 = "".join(map(str, out)) == vec["out"] and rv1 == vec["rv"]
            match["generic"] += is0
            match["acc"] += is1
            if not (is0 or is1) and first_bad is None:
                first_bad = (vi, vec["kind"], vec["rv"], rv0)
        n = len(code["vectors"])
        assert first_bad is None, (f"{code['name']} ({code['site']}): vector {first_bad[0]} ({first_bad[1]}) is neither the generic "
                                   f"nor the accelerated decoder's output (rv {first_bad[2]}, generic rv {first_bad[3]}); "
                                   f"{match['generic']} / {match['acc']} of {n} match D1 / D1b")
        if match["generic"] == n and not (acc_ok and match["acc"] == n):
            verdict = "generic"
        elif match["acc"] == n and match["generic"] < n:
            verdict = "acc"
        elif match["generic"] == n and match["acc"] == n:
            verdict = "either"          # the vectors do not separate the two decoders (does not happen with the kit's inputs)
        else:
            raise AssertionError(f"{code['name']}: vectors split between the decoders: {match} of {n}")
        report[code["name"]] = verdict
    choosable = {v for k, v in report.items() if v != "either" and
                 any(c["name"] == k and c["K"] in (5, 7) and 2 <= c["N"] <= 4 for c in pins["conv"])}
    assert len(choosable) <= 1, f"the K = 5 / 7 codes disagree on the decoder: {report}"
    for c in pins["conv"]:
        if not (c["K"] in (5, 7) and 2 <= c["N"] <= 4):
            assert report[c["name"]] == "generic", f"{c['name']} can only run the generic decoder: {report}"
    report["decoder"] = choosable.pop() if choosable else "undetermined"
    return report


def check_dsp(pins, orc):
    lib = orc.lib()
    d = pins["dsp"]
    rep = {}
    worst = 0.0
    for t in d["sig_normalize"]:
        x, want = _cv(t["in"]), _cv(t["out"])
        out = np.zeros(x.size // t["decim"] + 1, np.complex64)
        lib.orc_sig_normalize.restype = C.c_int
        n = lib.orc_sig_normalize(_ptr(x), C.c_int(x.size), C.c_int(t["decim"]), C.c_float(t["freq_shift"]), _ptr(out))
        assert n == want.size, ("sig_normalize length", n, want.size)
        err = float(np.max(np.abs(out[:n] - want)))
        worst = max(worst, err)
        assert err < 5e-6 * max(1.0, float(np.max(np.abs(want)))), ("sig_normalize (decision D2)", t["decim"], t["freq_shift"], err)
    rep["sig_normalize_max_err"] = worst
    for t in d["correlate"]:
        f, g, want = _cv(t["f"]), _cv(t["g"]), _cv(t["out"])
        out = np.zeros(g.size, np.complex64)
        lib.orc_correlate.restype = C.c_int
        n = lib.orc_correlate(_ptr(f), C.c_int(f.size), _ptr(g), C.c_int(g.size), C.c_int(t["step"]), _ptr(out))
        assert n == want.size, ("correlate length", n, want.size)
        assert np.max(np.abs(out[:n] - want)) < 2e-5 * np.max(np.abs(want)), ("correlate", t["step"])
    lib.orc_peak_energy_find.restype = C.c_float
    exact = 0
    for t in d["peak_energy_find"]:
        cv = _cv(t["cv"])
        pv = Cf()
        alg = 2 if t["alg"] == "early_late" else 0
        pos = lib.orc_peak_energy_find(_ptr(cv), C.c_int(cv.size), C.c_int(t["win"]), C.c_int(alg), C.byref(pv))
        if alg == 2:
            assert pos == np.float32(t["pos"]), ("peak_energy_find early / late (decision D3)", pos, t["pos"])
            assert abs(complex(pv.re, pv.im) - complex(*t["peak"])) < 2e-5 * max(1.0, abs(complex(*t["peak"]))), ("peak value", t["pos"])
            exact += 1
        else:
            assert abs(pos - t["pos"]) < 1e-5 * max(1.0, abs(t["pos"])), ("peak_energy_find weighted window", pos, t["pos"])
    rep["early_late_positions_identical"] = exact
    for t in d["peaks_scan"]:
        cv = _cv(t["cv"])
        idx = (C.c_int * 6)()
        lib.orc_peaks_scan(_ptr(cv), C.c_int(cv.size), idx, C.c_int(6))
        assert list(idx) == t["idx"], ("peaks_scan", list(idx), t["idx"])
    for t in d["rotate"]:
        x, want = _cv(t["in"]), _cv(t["out"])
        lib.orc_rotate(_ptr(x), C.c_int(x.size), C.c_float(t["rps"]))
        assert np.max(np.abs(x - want)) < 5e-6 * np.max(np.abs(want)), "rotate"
    lib.orc_interpolate_point.restype = Cf
    for t in d["interpolate_point"]:
        cv = _cv(t["cv"])
        for at, val in zip(t["at"], t["val"]):
            v = lib.orc_interpolate_point(_ptr(cv), C.c_int(cv.size), C.c_float(at))
            assert abs(complex(v.re, v.im) - complex(*val)) < 2e-5 * max(1.0, abs(complex(*val))), ("interpolate_point (decision D3b)", at)
    lib.orc_sinc.restype = C.c_float
    for x, y in zip(d["sinc"]["x"], d["sinc"]["y"]):
        assert abs(lib.orc_sinc(C.c_float(x)) - y) < 1e-6, ("sinc", x)
    for t in d["convolve_no_delay"]:
        f, g, want = _cv(t["f"]), _cv(t["g"]), _cv(t["out"])
        taps = np.ascontiguousarray(f.real, np.float32)
        out = np.zeros(g.size, np.complex64)
        lib.orc_convolve_nodelay_real(_ptr(taps), C.c_int(taps.size), _ptr(g), C.c_int(g.size), _ptr(out))
        assert want.size == g.size and np.max(np.abs(out - want)) < 2e-5 * np.max(np.abs(want)), "convolve (CONV_NO_DELAY)"
    return rep


def check(pins, orc):
    assert pins.get("format") == 1 and pins.get("end") is True, "not a complete pin_3p.c output"
    rep = {"library": pins["library"]}
    rep.update(check_conv(pins, orc))
    rep.update(check_dsp(pins, orc))
    return rep
