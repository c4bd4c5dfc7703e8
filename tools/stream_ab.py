#!/usr/bin/env python3
"""A/B of the stream engine's host side (csrc/stream_core.h) through fun_amd::receiver_chain::process_samples in device mode:
helper count x placement (FOA_STREAM_AFFINITY) x narrowing loop (FOA_STREAM_NO_AVX512) x batch size, one JSON line per run.
usage (GPU box): tools/stream_ab.py [frames] > gpurun_out/stream_ab.jsonl"""
import json
import os
import re
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fun_ofdm_amd as foa                      # noqa: E402
from fun_ofdm_amd import synth                  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
rx = foa.Receiver(0)
pays = synth.splitmix64_bytes(0xB57, n, 1024)
frames = rx.tx_build_frames(torch.from_numpy(pays).to("cuda:0"), 10)
s = frames.shape[1]
iq = rx.tx_channel(frames, s + 160, 80, 25.0, seed=5).cpu().numpy().reshape(-1).view(np.complex64)
rx.close()
cap, exe = "/tmp/stream.fc32", "/tmp/foa_sim"
iq.tofile(cap)
libdir = os.path.dirname(foa.library_path())
subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "examples", "foa_sim.cpp"), "-I", os.path.join(ROOT, "include"), "-L", libdir,
                "-lfun_ofdm_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", exe], check=True)
print(json.dumps({"lscpu": subprocess.run("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA|L3'", shell=True, capture_output=True, text=True).stdout}), flush=True)
runs = []
for batch in (1 << 22, 1 << 23):
    for helpers in (4, 6, 8, 12):
        for env in ({}, {"FOA_STREAM_AFFINITY": "1"}):
            if helpers == 4 and env:
                continue
            runs.append((batch, helpers, env))
for batch, helpers, env in runs:
    best = None
    for rep in range(2):
        r = subprocess.run([exe, cap, "--format", "fc32", "--preload", "--chunk", "4096", "--device-batch", str(batch), "--narrow-threads", str(helpers)],
                           capture_output=True, text=True, env=dict(os.environ, FOA_STREAM_STATS="1", **env))
        m = re.search(r"([\d.]+) Msamples/s through process_samples", r.stdout)
        p = re.search(r"(\d+) packets", r.stdout)
        if m and (best is None or float(m.group(1)) > best[0]):
            best = (float(m.group(1)), int(p.group(1)) if p else None, [ln for ln in r.stderr.splitlines() if ln.startswith("foa_stream:")])
    print(json.dumps({"batch": batch, "helpers": helpers, "env": env, "Msamples_per_s": best[0] if best else None, "packets": best[1] if best else None,
                      "stats": best[2] if best else r.stderr[-300:]}), flush=True)
