#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc passes of tools/profile_round.sh to per-launch, per-kernel figures (HBM bytes, SQ issue, LDS, L2)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
frames = 10000
args = sys.argv[3:]
if "--frames" in args:
    frames = int(args[args.index("--frames") + 1])


def per_kernel(sub, counter):
    acc = defaultdict(lambda: [0.0, set()])
    for path in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                low = {k.lower(): v for k, v in row.items()}
                if low.get("counter_name") != counter:
                    continue
                name = low["kernel_name"].split("(")[0].split("<")[0].split("::")[-1].strip()       # (template arguments dropped: k_header<float2> -> k_header)
                acc[name][0] += float(low["counter_value"])
                acc[name][1].add(low.get("dispatch_id"))
    return {k: v[0] / max(1, len(v[1])) for k, v in acc.items()}


fetch, write = per_kernel("pf", "FETCH_SIZE"), per_kernel("pw", "WRITE_SIZE")
kern = {}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    kern[k] = {"FETCH_SIZE_KB_per_launch": round(f, 1), "WRITE_SIZE_KB_per_launch": round(w, 1),
               "hbm_bytes_per_launch": int((2.0 * f + w) * 1024)}
doc = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline " + " ".join(args),
       "frames_per_gpu": frames,
       "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B; MI355X_MICROARCH.md HBM section)",
       "kernels": kern}
json.dump(doc, open(os.path.join(out, tag + "_pmc_hbm.json"), "w"), indent=1)
sq = {}
for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAVES"):
    for k, v in per_kernel("ps", c).items():
        if k.startswith("k_"):
            sq.setdefault(k, {})[c] = round(v, 1)
if sq:
    json.dump({"command": "rocprofv3 --pmc SQ_* (own pass) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline " + " ".join(args),
               "frames_per_gpu": frames, "per_launch": sq}, open(os.path.join(out, tag + "_pmc_sq.json"), "w"), indent=1)
lds = {}
for sub, names in (("pl", ("SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL")),
                   ("pc", ("TCC_HIT_sum", "TCC_MISS_sum"))):
    for c in names:
        for k, v in per_kernel(sub, c).items():
            if k.startswith("k_"):
                lds.setdefault(k, {})[c] = round(v, 1)
for k, d in lds.items():
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"], 4)
    if "TCC_HIT_sum" in d and d["TCC_HIT_sum"] + d.get("TCC_MISS_sum", 0.0) > 0:
        d["l2_hit_frac"] = round(d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), 4)
if lds:
    json.dump({"command": "rocprofv3 --pmc SQ_LDS_* | TCC_HIT_sum TCC_MISS_sum (own passes) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline " + " ".join(args),
               "frames_per_gpu": frames,
               "notes": "lds_bank_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (extra cycles over all LDS-array cycles); l2_hit_frac = TCC_HIT / (TCC_HIT + TCC_MISS), MI355X_MICROARCH.md L2 section",
               "per_launch": lds}, open(os.path.join(out, tag + "_pmc_lds_l2.json"), "w"), indent=1)
# effective clock: GRBM_GUI_ACTIVE / launch duration (MI355X_MICROARCH.md "DVFS give-back"); the counter pass serialises kernels
clk = {}
for path in glob.glob(os.path.join(out, "pg", "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            low = {k.lower(): v for k, v in row.items()}
            if low.get("counter_name") != "GRBM_GUI_ACTIVE" or "start_timestamp" not in low:
                continue
            name = low["kernel_name"].split("(")[0].split("<")[0].split("::")[-1].strip()       # (template arguments dropped: k_header<float2> -> k_header)
            if not name.startswith("k_"):
                continue
            dur = float(low["end_timestamp"]) - float(low["start_timestamp"])
            a = clk.setdefault(name, [0.0, 0.0, 0])
            a[0] += float(low["counter_value"]); a[1] += dur; a[2] += 1
if clk:
    doc = {"command": "rocprofv3 --pmc GRBM_GUI_ACTIVE (own pass) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --no-sync-leg " + " ".join(args),
           "frames_per_gpu": frames,
           "notes": "ghz = GRBM_GUI_ACTIVE / (End_Timestamp - Start_Timestamp) of the same dispatches; if the counter is summed over the 8 XCDs the "
                    "figure is 8 x the clock (ghz_if_summed_over_8_xcd)",
           "per_kernel": {k: {"launches": v[2], "GRBM_GUI_ACTIVE_per_launch": round(v[0] / v[2], 1), "duration_us_per_launch": round(v[1] / v[2] / 1e3, 2),
                              "ghz": round(v[0] / v[1], 3), "ghz_if_summed_over_8_xcd": round(v[0] / v[1] / 8, 3)} for k, v in clk.items() if v[1] > 0}}
    json.dump(doc, open(os.path.join(out, tag + "_pmc_clock.json"), "w"), indent=1)
print(json.dumps(kern, indent=1))
