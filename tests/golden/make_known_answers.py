#!/usr/bin/env python3
"""Derives tests/golden/known_answers.json.

The reference has no tests or vectors and cannot be built or imported here, so
these are the known answers that follow from its in-tree code alone (SURVEY.md
Appendix D2), re-derived by this script from the formulas the reference states:

  scrambler      reference src/l1/scramb.c:39-52  (reg 0x4d4b, b = (r>>14 ^ r)&1, r = r<<1|b)
  deinterleave   reference src/l1/interleave.c:81-86  (kep = N*((5*kc)&7) + (kc>>3), N=53)
  conv_spot      reference src/l1/conv.c: generator polynomials from the comments at
                 :123-128 (k5_12), :148-154 (k5_13), :174-181 (k5_14), :201-209 (k5_15), :518-523 (tch3)
                 and rows of the next_output tables at :130-135, :155-160, :183-188, :210-215, :525-542
                 read off the file; k9_13: polynomials :345-351, rows of the table at :353-419
  xch_dc12       reference src/l1/xch_dc12.c:45-54 + src/l1/punct.c:1105-1125: first punctured positions
                 of P(12;13) = mask 110 101 011 x4 + 111 (0 = punctured)
  tch9_punct     reference src/l1/tch9.c:72-78 + src/l1/punct.c:48-175: the 320 punctured positions of
                 the 9k6 mode
  fcch           reference src/sdr/fcch.c:600-613 constants
"""
import json
import os

r = 0x4D4B
bits = []
for _ in range(432):
    b = ((r >> 14) ^ r) & 1
    r = ((r << 1) | b) & 0xFFFF
    bits.append(b)

kc = list(range(16))
out = {
    "scrambler": {
        "first64": "".join(map(str, bits[:64])),
        "ones_in_first": {str(n): sum(bits[:n]) for n in (96, 208, 424, 432)},
    },
    "deinterleave_intra_53_first16": [53 * ((5 * k) & 7) + (k >> 3) for k in kc],
    # rows {state: [out(b=0), out(b=1)]} as printed in the reference tables
    "conv_spot": {
        "k5_12": {"K": 5, "polys": [0x19, 0x17],
                  "next_output_rows": {"0": [0, 3], "1": [1, 2], "4": [2, 1], "8": [3, 0], "15": [1, 2]}},
        "k5_14": {"K": 5, "polys": [0x19, 0x17, 0x15, 0x1F],
                  "next_output_rows": {"0": [0, 15], "1": [5, 10], "4": [9, 6], "8": [15, 0], "15": [4, 11]}},
        "k5_13": {"K": 5, "polys": [0x15, 0x1B, 0x1F],
                  "next_output_rows": {"0": [0, 7], "1": [3, 4], "2": [5, 2], "7": [5, 2], "8": [7, 0], "15": [2, 5]}},
        "k5_15": {"K": 5, "polys": [0x15, 0x1B, 0x1F, 0x1D, 0x17],
                  "next_output_rows": {"0": [0, 31], "1": [13, 18], "3": [26, 5], "4": [14, 17], "8": [31, 0],
                                       "15": [11, 20]}},
        "k9_13": {"K": 9, "polys": [0x1ED, 0x19B, 0x127],
                  "next_output_rows": {"0": [0, 7], "1": [3, 4], "2": [5, 2], "3": [6, 1], "7": [0, 7],
                                       "100": [4, 3], "128": [7, 0], "255": [2, 5]}},
        "tch3_k7": {"K": 7, "polys": [0x6D, 0x4F],
                    "next_output_rows": {"0": [0, 3], "1": [1, 2], "2": [3, 0], "16": [2, 1], "32": [3, 0],
                                         "63": [0, 3]}},
    },
    "fcch": {"bin_hz": 200.0, "chirp_rate_hz_per_ms": 2995.2, "bcch_period_symbols": 7488},
    "xch_dc12_punctured_first13": [2, 4, 6, 11, 13, 15, 20, 22, 24, 29, 31, 33, 41],
    "tch9_9k6_punctured": sorted([1, 5] + [10 + 6 * k for k in range(158)] + [13 + 6 * k for k in range(158)] + [963, 967]),
}
# SURVEY.md Appendix D2 lists these; assert the derivation agrees with what the survey recorded
assert out["scrambler"]["first64"] == "0001001100011011110001000010010100001111100011000001010111101111"
assert out["scrambler"]["ones_in_first"] == {"96": 48, "208": 101, "424": 211, "432": 216}
assert out["deinterleave_intra_53_first16"] == [0, 265, 106, 371, 212, 53, 318, 159, 1, 266, 107, 372, 213, 54, 319, 160]

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "known_answers.json"), "w") as f:
    json.dump(out, f, indent=1)
print("written")
