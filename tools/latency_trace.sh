#!/bin/bash
# usage (GPU box): tools/latency_trace.sh [batch samples] [frames]  -- process_samples in device mode at the air's pace under a rocprofv3 kernel
# trace: what the GPU does, queue by queue, between a small batch's upload and its payloads (tools/trace_timeline.py prints a window of it)
B=${1:-4096}; n=${2:-3000}
root=$PWD
python3 - "$n" <<'PY'
import os, subprocess, sys
import numpy as np, torch
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
import fun_ofdm_amd as foa
from fun_ofdm_amd import synth
n = int(sys.argv[1])
rx = foa.Receiver(0)
pays = synth.splitmix64_bytes(0xB57, n, 1024)
pays[:, :4] = np.arange(n, dtype="<u4").view(np.uint8).reshape(n, 4)
frames = rx.tx_build_frames(torch.from_numpy(pays).to("cuda:0"), 10)
s = frames.shape[1]
iq = rx.tx_channel(frames, s + 160, 80, 25.0, seed=5).cpu().numpy().reshape(-1).view(np.complex64)
rx.close()
iq.tofile("/tmp/stream_lat.fc32")
libdir = os.path.dirname(foa.library_path())
subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "examples", "foa_sim.cpp"), "-I", os.path.join(ROOT, "include"), "-L", libdir,
                "-lfun_ofdm_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", "/tmp/foa_sim_lat"], check=True)
print(s)
PY
cd /tmp && export TMPDIR=/tmp && cd $root
rm -rf gpurun_out/lat_trace
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/lat_trace -o lt -- /tmp/foa_sim_lat /tmp/stream_lat.fc32 --format fc32 --preload --chunk 4096 --device-batch $B --narrow-threads 8 --pace 20 2>&1 | tail -3
python3 tools/trace_timeline.py $(find gpurun_out/lat_trace -name "*kernel_trace.csv" | head -1) 0.6 1800
