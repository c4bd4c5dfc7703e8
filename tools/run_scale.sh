#!/bin/bash
# usage (on a multi-GPU node, from the repo root):  tools/run_scale.sh [steps] [warmup]
# BASELINE config 4's curve: bench.py at N = 1, 2, 4, 8 ranks (as many as the node has devices), each started the way the driver starts it --
# python -m torch.distributed.run, one rank per GPU over RCCL -- back to back; one JSON line per N into gpurun_out/scale_N.json, the
# efficiency table (value_N / (N x value_1): weak scaling, 10 000 frames per GPU) on stdout, and two checks that need no edit at N = 8:
#   * every line's collective reports N ranks over RCCL and bit-exact PSDUs (exit status 1 otherwise);
#   * one stream over the node's devices behind the C ABI: examples/foa_sim --devices 0,..,N-1 (foa_shard_*: batch k on device k mod N) must
#     write the same payload records as the one-device run (exit status 1 otherwise).
# On a one-GPU box this is the N = 1 line and the shard engine with its one device listed once and twice.
steps=${1:-20}; warmup=${2:-5}
ndev=$(python3 -c "import torch; print(torch.cuda.device_count())")
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 1 2 4 8; do
  [ "$n" -gt "$ndev" ] && break
  port=$((29500 + n))
  if [ "$n" = 1 ]; then FOA_BENCH_FORCE_DIST=1 python3 bench.py --gpus 1 --steps $steps --warmup $warmup --no-extra-legs > gpurun_out/scale_$n.json 2> gpurun_out/scale_$n.err
  else python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port bench.py --gpus $n --steps $steps --warmup $warmup \
         --no-extra-legs > gpurun_out/scale_$n.json 2> gpurun_out/scale_$n.err; fi
done
python3 - <<'PY' || exit 1
import glob, json, sys
rows, bad = [], 0
for f in sorted(glob.glob("gpurun_out/scale_*.json"), key=lambda p: int(p.split("_")[-1].split(".")[0])):
    try:
        d = json.loads([ln for ln in open(f) if ln.startswith("{")][0])
        c = d["config"].get("collective", {})
        rows.append((d["n_gpus"], d["value"], d["ms_per_step"], d["config"]["psdu_bit_exact"], c.get("backend"), c.get("ranks")))
    except Exception as e:
        print(f, "unreadable:", e); bad += 1
if rows:
    v1 = rows[0][1] / rows[0][0]
    print("%4s %14s %10s %10s %9s %s" % ("GPUs", "Msamples/s", "ms/step", "efficiency", "bit-exact", "collective"))
    for n, v, ms, ok, be, ranks in rows:
        print("%4d %14.1f %10.4f %10.3f %9s %s x %s" % (n, v, ms, v / (n * v1), ok, be or "-", ranks))
        if not ok or be != "nccl" or ranks != n:
            print("  ^ expected bit-exact PSDUs and an RCCL collective of %d ranks" % n); bad += 1
sys.exit(1 if bad or not rows else 0)
PY
# ---- one stream over the node's devices (foa_shard_*), against the one-device payload records
python3 - "$ndev" <<'PY' || exit 1
import os, subprocess, sys
import numpy as np, torch
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
import fun_ofdm_amd as foa
from fun_ofdm_amd import synth
ndev = int(sys.argv[1])
rx = foa.Receiver(0)
n = 6000
pays = synth.splitmix64_bytes(0xB57, n, 1024)
frames = rx.tx_build_frames(torch.from_numpy(pays).to("cuda:0"), 10)
s = frames.shape[1]
iq = rx.tx_channel(frames, s + 160, 80, 25.0, seed=5).cpu().numpy().reshape(-1).view(np.complex64)
rx.close()
cap, exe = "/tmp/scale_stream.fc32", "/tmp/foa_sim_scale"
iq.tofile(cap)
libdir = os.path.dirname(foa.library_path())
subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "examples", "foa_sim.cpp"), "-I", os.path.join(ROOT, "include"), "-L", libdir,
                "-lfun_ofdm_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", exe], check=True)
def records(devs):
    out = "/tmp/scale_%s.rec" % devs.replace(",", "_")
    r = subprocess.run([exe, cap, "--format", "fc32", "--preload", "--chunk", "4096", "--device-batch", str(1 << 18), "--narrow-threads", "4", "--devices", devs, "--out", out],
                       capture_output=True, text=True, timeout=600)
    if r.returncode:
        print(r.stdout[-400:], r.stderr[-400:]); sys.exit(1)
    return open(out, "rb").read()
want = records("0")
lists = ["0,0"] if ndev == 1 else [",".join(str(d) for d in range(k)) for k in (2, 4, 8) if k <= ndev]
bad = 0
for devs in lists:
    same = records(devs) == want
    print("foa_sim --devices %-16s %d bytes of payload records, same as one device: %s" % (devs, len(want), same))
    bad += not same
sys.exit(1 if bad or len(want) < n * 1000 else 0)
PY
