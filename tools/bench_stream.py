#!/usr/bin/env python3
"""Throughput of the drop-in API itself: a capture of back-to-back 54 Mbps frames through
fun_amd::receiver_chain::process_samples (examples/foa_sim.cpp --preload: the complex<double> chunks are prepared first and only
the receive loop is timed), in its three modes: synchronous, asynchronous batches over the host pre-sync, and everything on the
device (foa_stream_*).  Prints one JSON line per run; Msamples/s against the 20 Msample/s of the air.
usage: tools/bench_stream.py [frames] > gpurun_out/stream.jsonl"""
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
iq = rx.tx_channel(frames, s + 160, 80, 25.0, seed=5).cpu().numpy().reshape(-1).view(np.complex64)      # 8 us between frames
rx.close()
cap = "/tmp/stream.fc32"
iq.tofile(cap)
exe = "/tmp/foa_sim"
libdir = os.path.dirname(foa.library_path())
subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "examples", "foa_sim.cpp"), "-I", os.path.join(ROOT, "include"), "-L", libdir,
                "-lfun_ofdm_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", exe], check=True)
short = min(n, 12000)
runs = [("synchronous", ["--chunk", "4096"], 600), ("async 8", ["--chunk", "4096", "--async", "8"], short), ("async 32", ["--chunk", "4096", "--async", "32"], short)]
for batch in (1 << 20, 1 << 22, 1 << 24):
    for chunk, threads in ((4096, 0), (4096, 4), (4096, 8), (4096, 16), (65536, 8), (1 << 20, 16)):
        if batch != 1 << 22 and (chunk, threads) not in ((4096, 0), (4096, 8)):
            continue
        runs.append(("device %dM, calls of %d, %d helpers" % (batch >> 20, chunk, threads),
                     ["--chunk", str(chunk), "--device-batch", str(batch), "--narrow-threads", str(threads)], n))
if len(sys.argv) > 2 and sys.argv[2] == "quick":      # two device-mode runs only, with the submitter's time split (FOA_STREAM_STATS)
    runs = [r for r in runs if r[0] in ("device 4M, calls of 4096, 0 helpers", "device 4M, calls of 4096, 4 helpers", "device 16M, calls of 4096, 8 helpers")]
    os.environ["FOA_STREAM_STATS"] = "1"
for name, extra, frames_used in runs:
    src = cap
    if frames_used < n:                        # the synchronous mode needs a millisecond per call: a shorter capture
        src = "/tmp/stream_%d.fc32" % frames_used
        iq[:frames_used * (s + 160)].tofile(src)
    pre = os.environ.get("FOA_SIM_PREFIX", "").split()      # e.g. "taskset -c 0-7": where the chain's threads may run
    r = subprocess.run(pre + [exe, src, "--format", "fc32", "--preload"] + extra, capture_output=True, text=True)
    m = re.search(r"([\d.]+) Msamples/s through process_samples \((\d+) samples in ([\d.]+) s, (\d+) calls of (\d+)\)", r.stdout)
    p = re.search(r"(\d+) packets", r.stdout)
    for line in r.stderr.splitlines():
        if line.startswith("foa_stream:"):
            print(json.dumps({"mode": name, "stats": line}), flush=True)
    if not m:
        print(json.dumps({"mode": name, "error": (r.stdout + r.stderr)[-400:]}), flush=True)
        continue
    rate = float(m.group(1))
    print(json.dumps({"mode": name, "args": extra, "Msamples_per_s": rate, "x_realtime_20MSps": round(rate / 20.0, 1), "samples": int(m.group(2)),
                      "seconds": float(m.group(3)), "calls": int(m.group(4)), "chunk": int(m.group(5)), "packets": int(p.group(1)) if p else None,
                      "frames_sent": frames_used}), flush=True)
