"""The pinning kit (tools/pin_3p.c, tests/pin_check.py) works before anybody needs it.

tools/pin_3p.c is written against the REAL libosmocore / libosmo-dsp headers, which this image lacks.  Here it is
(1) compiled as strict C99 against declaration-only headers carrying the two libraries' public prototypes, and
(2) linked against the oracle standing in for the libraries (tests/c/pin_3p_oracle_shim.c), run, and its JSON pushed
through the checker: with the stand-in behaving like an old libosmocore every code must come out "generic", with the
accelerated decoder switched on the K = 5 / 7 codes with N <= 4 must come out "acc" and the N = 5 and K = 9 codes
"generic".  This pins nothing -- it proves the kit's plumbing and that the checker separates the decoders on the kit's
vectors.  The real thing: INTEGRATION.md, "Pinning the third-party arithmetic"."""
import json
import os
import subprocess
import textwrap

import pytest

import pin_check

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# the public declarations tools/pin_3p.c uses, as libosmocore's <osmocom/core/{bits,conv}.h> and libosmo-dsp's
# <osmocom/dsp/{cxvec,cxvec_math}.h> publish them
HEADERS = {
    "osmocom/core/bits.h": """
        #include <stdint.h>
        typedef int8_t sbit_t; typedef uint8_t ubit_t; typedef uint8_t pbit_t;
    """,
    "osmocom/core/conv.h": """
        #include <stdint.h>
        #include <osmocom/core/bits.h>
        enum osmo_conv_term { CONV_TERM_FLUSH = 0, CONV_TERM_TRUNCATION, CONV_TERM_TAIL_BITING };
        struct osmo_conv_code {
            int N; int K; int len;
            enum osmo_conv_term term;
            const uint8_t (*next_output)[2];
            const uint8_t (*next_state)[2];
            const uint8_t *next_term_output;
            const uint8_t *next_term_state;
            const int *puncture;
        };
        int osmo_conv_get_output_length(const struct osmo_conv_code *code, int len);
        int osmo_conv_decode(const struct osmo_conv_code *code, const sbit_t *input, ubit_t *output);
    """,
    "osmocom/dsp/cxvec.h": """
        #include <complex.h>
        #define CXVEC_FLG_REAL_ONLY (1 << 0)
        struct osmo_cxvec { int len, max_len, flags; float complex *data; float complex _data[0]; };
        struct osmo_cxvec *osmo_cxvec_alloc(int max_len);
        void osmo_cxvec_free(struct osmo_cxvec *cv);
    """,
    "osmocom/dsp/cxvec_math.h": """
        #include <complex.h>
        #include <osmocom/dsp/cxvec.h>
        enum osmo_cxvec_conv_type { CONV_FULL_SPAN, CONV_OVERLAP_ONLY, CONV_NO_DELAY };
        enum osmo_cxvec_peak_alg { PEAK_WEIGH_WIN, PEAK_WEIGH_WIN_CENTER, PEAK_EARLY_LATE };
        float osmo_sinc(float x);
        struct osmo_cxvec *osmo_cxvec_rotate(const struct osmo_cxvec *in, float rps, struct osmo_cxvec *out);
        struct osmo_cxvec *osmo_cxvec_convolve(const struct osmo_cxvec *f, const struct osmo_cxvec *g,
                                               enum osmo_cxvec_conv_type type, struct osmo_cxvec *out);
        struct osmo_cxvec *osmo_cxvec_correlate(const struct osmo_cxvec *f, const struct osmo_cxvec *g, int g_corr_step,
                                                struct osmo_cxvec *out);
        float complex osmo_cxvec_interpolate_point(const struct osmo_cxvec *cv, float pos);
        float osmo_cxvec_peak_energy_find(const struct osmo_cxvec *cv, int win_size, enum osmo_cxvec_peak_alg alg,
                                          float complex *peak_val_p);
        void osmo_cxvec_peaks_scan(const struct osmo_cxvec *cv, int *peaks_idx, int N);
        struct osmo_cxvec *osmo_cxvec_sig_normalize(const struct osmo_cxvec *sig, int decim, float freq_shift,
                                                    struct osmo_cxvec *out);
    """,
}


@pytest.fixture(scope="module")
def kit(tmp_path_factory, orc):
    tmp = tmp_path_factory.mktemp("pin_kit")
    for rel, txt in HEADERS.items():
        p = tmp / "tp" / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        guard = "PINKIT_" + rel.replace("/", "_").replace(".", "_").upper()
        p.write_text(f"#ifndef {guard}\n#define {guard}\n{textwrap.dedent(txt)}\n#endif\n")
    inc = ["-I" + str(tmp / "tp")]
    src = os.path.join(ROOT, "tools", "pin_3p.c")
    # (1) C99, every warning an error (not -pedantic: libosmo-dsp's own struct ends in a zero-length array)
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-O2", "-c", src, "-o", str(tmp / "pin_3p.o"),
                        "-DPIN_3P_LIBRARY=\"oracle self-test\""] + inc, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # (2) link against the oracle standing in for the two libraries
    liborc = orc.build()
    exe = str(tmp / "pin_3p")
    r = subprocess.run(["gcc", "-std=gnu99", "-O2", "-o", exe, str(tmp / "pin_3p.o"), os.path.join(ROOT, "tests", "c", "pin_3p_oracle_shim.c"),
                        "-I" + os.path.join(ROOT, "oracle"), "-L" + os.path.dirname(liborc), "-l:" + os.path.basename(liborc),
                        "-Wl,-rpath," + os.path.dirname(liborc), "-lm"] + inc, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def _run(exe, mode):
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, PIN_SHIM_CONV_MODE=str(mode)))
    assert r.returncode == 0, r.stderr
    return json.loads(r.stdout)


def test_kit_output_against_a_generic_only_library(kit, orc):
    pins = _run(kit, 0)
    rep = pin_check.check(pins, orc)
    assert rep["library"] == "oracle self-test"
    assert rep["decoder"] == "generic"
    assert all(rep[c["name"]] == "generic" for c in pins["conv"])
    assert len(pins["conv"]) == 9 and all(len(c["vectors"]) == 28 for c in pins["conv"])
    assert rep["early_late_positions_identical"] == 12


def test_kit_output_against_a_library_with_the_accelerated_decoder(kit, orc):
    pins = _run(kit, 1)
    rep = pin_check.check(pins, orc)
    assert rep["decoder"] == "acc"
    for c in pins["conv"]:
        want = "acc" if c["K"] in (5, 7) and c["N"] <= 4 else "generic"
        assert rep[c["name"]] == want, (c["name"], rep)


def test_the_kit_inputs_are_the_same_on_every_run(kit):
    a, b = _run(kit, 0), _run(kit, 0)
    assert a == b


def test_checker_refuses_a_wrong_library(kit, orc):
    pins = _run(kit, 0)
    v = pins["conv"][0]["vectors"][3]
    v["out"] = ("1" if v["out"][0] == "0" else "0") + v["out"][1:]
    with pytest.raises(AssertionError):
        pin_check.check(pins, orc)
    pins = _run(kit, 0)
    pins["dsp"]["peak_energy_find"][1]["pos"] += 1.0 / 512.0
    with pytest.raises(AssertionError):
        pin_check.check(pins, orc)
