#!/usr/bin/env python3
"""profiles/valu_k_rx4.json from a tools/pmc_rx4.sh summary: what the headline kernel's vector ALUs were doing, keyed on the
hash of the sources the kernel is compiled from (bench.py reports it as `roofline_valu` only while that hash matches).
    python3 tools/valu_summary.py profiles/r05m_pmc_sq_k_rx4_interleaved.txt r05m"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

src, tag = sys.argv[1], sys.argv[2]
per_wave = {}
for line in open(src):
    m = re.match(r"\s+(SQ_\w+)\s+[\d.]+\s+per wave\s+([\d.]+)", line)
    if m:
        per_wave[m.group(1)] = float(m.group(2))
waves_per_simd = 6                       # kRx4Waves<16, 4>: amdgpu_waves_per_eu(6, 6)
busy = per_wave["SQ_ACTIVE_INST_VALU"] * waves_per_simd / per_wave["SQ_WAVE_CYCLES"]
out = {"_comment": "SQ counters of k_rx4<16,4> over 100000 bursts (tools/pmc_rx4.sh, two passes of eight counters), per wave of four "
                   "bursts; valu_busy = SQ_ACTIVE_INST_VALU x resident waves per SIMD / SQ_WAVE_CYCLES: the share of the time a "
                   "SIMD's vector ALU is executing an instruction of one of its six waves",
       "tag": tag, "source": os.path.relpath(src, ROOT), "waves_per_simd": waves_per_simd,
       "valu_insts_per_wave": per_wave["SQ_INSTS_VALU"], "salu_insts_per_wave": per_wave.get("SQ_INSTS_SALU"),
       "lds_insts_per_wave": per_wave.get("SQ_INSTS_LDS"), "active_inst_valu_per_wave": per_wave["SQ_ACTIVE_INST_VALU"],
       "wave_cycles_per_wave": per_wave["SQ_WAVE_CYCLES"], "valu_busy": busy,
       "kernel_sources_sha256": bench.kernel_sources_hash()}
json.dump(out, open(os.path.join(ROOT, "profiles", "valu_k_rx4.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
