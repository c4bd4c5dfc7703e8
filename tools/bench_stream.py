#!/usr/bin/env python3
"""Single-stream real-time check of the C++ front end: a capture of back-to-back 54 Mbps frames through
examples/foa_sim.cpp (file_source -> receiver -> receiver_chain -> callback), synchronous and in asynchronous batches.
Prints samples per second against the 20 Msample/s of the air."""
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fun_ofdm_amd as foa                      # noqa: E402
from fun_ofdm_amd import synth                  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
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
air = iq.size / 20e6
for extra in ([], ["--async", "8"], ["--async", "32"]):
    t0 = time.perf_counter()
    r = subprocess.run([exe, cap, "--format", "fc32"] + extra, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    print("%-14s %s  | %.2f s for %.2f s of air = %.1f Msamples/s (%.2f x real time; includes process start-up)"
          % (" ".join(extra) or "synchronous", r.stdout.strip(), dt, air, iq.size / dt / 1e6, air / dt), flush=True)
