#!/bin/bash
# small batches through process_samples: loops in flight (FOA_STREAM_DEPTH) x batch size, with the submitter's time split
python3 - <<'PY'
import os, sys, subprocess, numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES","8")
import fun_ofdm_amd as foa
from fun_ofdm_amd import synth
n=30000
rx=foa.Receiver(0)
pays=synth.splitmix64_bytes(0xB57,n,1024)
frames=rx.tx_build_frames(torch.from_numpy(pays).to("cuda:0"),10)
s=frames.shape[1]
iq=rx.tx_channel(frames,s+160,80,25.0,seed=5).cpu().numpy().reshape(-1).view(np.complex64)
rx.close()
iq.tofile("/tmp/cap.fc32")
libdir=os.path.dirname(foa.library_path())
subprocess.run(["g++","-O2","-std=c++17","examples/foa_sim.cpp","-Iinclude","-L",libdir,"-lfun_ofdm_amd","-Wl,-rpath,"+libdir,"-Wl,-rpath,/opt/rocm/lib","-lpthread","-o","/tmp/foa_sim"],check=True)
os.environ["FOA_STREAM_STATS"]="1"
for B in (16384, 65536, 262144, 1<<20):
    for depth in (2, 3, 4):
        os.environ["FOA_STREAM_DEPTH"]=str(depth)
        r=subprocess.run(["/tmp/foa_sim","/tmp/cap.fc32","--format","fc32","--preload","--chunk","4096","--device-batch",str(B),"--narrow-threads","8"],capture_output=True,text=True)
        print("B",B,"depth",depth, r.stdout.splitlines()[0] if r.stdout else r.stderr[-300:])
        print("   ", [l for l in r.stderr.splitlines() if "submitter" in l][-1][-170:])
PY
