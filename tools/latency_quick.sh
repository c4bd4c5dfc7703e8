#!/bin/bash
# usage (GPU box): tools/latency_quick.sh <frames>  -- process_samples in device mode at the air's own pace (20 Msample/s) and at 20 x, default
# options (no stream_longest): payload latency percentiles per batch size.  A cut of tools/bench_latency.py for quick checks.
n=${1:-20000}
python3 - "$n" <<'PY'
import json, os, re, subprocess, sys
import numpy as np, torch
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
import fun_ofdm_amd as foa
from fun_ofdm_amd import synth
n = int(sys.argv[1])
rx = foa.Receiver(0)
pays = synth.splitmix64_bytes(0xB57, n, 1024)
pays[:, :4] = np.arange(n, dtype="<u4").view(np.uint8).reshape(n, 4)
frames = rx.tx_build_frames(torch.from_numpy(pays).to("cuda:0"), 10)
s = frames.shape[1]; PITCH, LEAD = s + 160, 80
iq = rx.tx_channel(frames, PITCH, LEAD, 25.0, seed=5).cpu().numpy().reshape(-1).view(np.complex64)
rx.close()
cap, exe = "/tmp/stream_lat.fc32", "/tmp/foa_sim_lat"
iq.tofile(cap)
libdir = os.path.dirname(foa.library_path())
subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "examples", "foa_sim.cpp"), "-I", os.path.join(ROOT, "include"), "-L", libdir,
                "-lfun_ofdm_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", exe], check=True)
CHUNK = os.environ.get("CHUNK", "4096")       # samples per process_samples call: payloads come back with a call, so this is the clock's tick
ROWS = ((1 << 12, 20), (1 << 13, 20), (1 << 14, 20), (1 << 14, 400), (1 << 16, 20), (1 << 16, 400), (1 << 18, 400), (1 << 22, 0))
for B, pace in ROWS[:int(os.environ.get("ROWS", len(ROWS)))]:
    extra = ["--chunk", CHUNK, "--device-batch", str(B), "--narrow-threads", "8"] + (["--pace", str(pace)] if pace else [])
    r = subprocess.run([exe, cap, "--format", "fc32", "--preload", "--latency", str(PITCH), str(LEAD), str(s)] + extra, capture_output=True, text=True, timeout=900)
    m = re.search(r"([\d.]+) Msamples/s through process_samples", r.stdout)
    lat = re.search(r"payload latency ms: p50 ([\d.]+) p90 ([\d.]+) p99 ([\d.]+) max ([\d.]+) \((\d+) payloads", r.stdout)
    p = re.search(r"(\d+) packets", r.stdout)
    print(json.dumps({"batch": B, "pace_Msps": pace, "Msamples_per_s": float(m.group(1)) if m else None, "chunk": int(CHUNK), "packets": int(p.group(1)) if p else None, "frames": n,
                      "latency_ms": dict(zip(("p50", "p90", "p99", "max"), map(float, lat.groups()[:4]))) if lat else None, "err": None if m else (r.stdout + r.stderr)[-300:]}), flush=True)
PY
