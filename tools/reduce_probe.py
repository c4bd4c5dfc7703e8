#!/usr/bin/env python3
"""tools/probe_issue.hip output -> profiles/<tag>_probe_issue.json (read by bench.py for the VALU issue roof).
usage: tools/reduce_probe.py gpurun_out/r02/probe_issue.txt profiles/r02_probe_issue.json"""
import json
import re
import sys

rows = {}
dev = None
for line in open(sys.argv[1]):
    m = re.match(r"(.+?)\s+W=(\d)\s+([\d.]+) ms\s+([\d.]+) clk per wave-instr per SIMD\s+\(a wave issues one per\s+([\d.]+) clk; slowest / fastest wave ([\d.]+)\)\s+([\d.]+) GHz", line)
    if m:
        rows.setdefault(m.group(1).strip(), {})["W%s" % m.group(2)] = {"clk_per_wave_instr_per_simd": float(m.group(4)), "own_issue_interval_clk": float(m.group(5)),
                                                                       "slowest_over_fastest_wave": float(m.group(6)), "ghz": float(m.group(7))}
    elif line.startswith("gfx"):
        dev = line.strip()
mix = rows.get("forward-step mix (8 VALU)") or rows["forward-step mix (9 VALU)"]
ghz = sorted(r["W5"]["ghz"] for r in rows.values() if "ds_read" not in str(r) and r["W5"]["ghz"] > 1.5)
doc = {"device": dev, "source": "tools/probe_issue.hip (fixed window, exactly W waves per SIMD), raw output next to this file",
       # the forward pass runs five waves per SIMD at 10 000 frames
       "forward_mix_clk_per_wave_instr": mix["W5"]["clk_per_wave_instr_per_simd"],
       "shader_clock_ghz": ghz[len(ghz) // 2],
       "reading": "packed-u16 / DPP / bfi / fma instructions issue once per 4.04-4.06 clocks per SIMD however many waves wait (v_add_u32 and "
                  "v_lshrrev_b32: 2.06; v_permlane32/16_swap: 8.05); the SIMD serves its OLDEST wave first, so with W >= 4 the youngest waves "
                  "get next to nothing (slowest / fastest wave -> 0) and a lone wave issues only every 6.3-7 clocks (13.4 for the swaps, 24 "
                  "through v_readfirstlane + SALU)",
       "rows": rows}
json.dump(doc, open(sys.argv[2], "w"), indent=1)
print(doc["forward_mix_clk_per_wave_instr"], doc["shader_clock_ghz"])
