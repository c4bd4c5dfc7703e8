"""Random streams through the C++ drop-in (fun_amd::receiver_chain::process_samples via examples/foa_sim.cpp) in its three modes --
synchronous, asynchronous batches over the host pre-sync, everything on the device -- with random chunk sizes, against the oracle's
receiver_chain in 4096-sample calls: ordered payload lists must be equal (the host pre-sync decides timing_sync.cpp:99 for the
reference's call size whatever the chunk size it is fed with).
Usage (GPU box, from the repo root): python3 tests/manual/stress_chain_cpp.py [first seed] [last seed]"""
import os
import subprocess
import sys
import tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import fun_ofdm_amd as foa
from fun_ofdm_amd import synth
from oracle import pyoracle as po

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tmp = tempfile.mkdtemp()
exe = os.path.join(tmp, "foa_sim")
libdir = os.path.dirname(foa.library_path())
subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "examples", "foa_sim.cpp"), "-I", os.path.join(ROOT, "include"),
                "-L", libdir, "-lfun_ofdm_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", exe], check=True)
bad = 0
runs = 0
for seed in range(lo, hi):
    rng = np.random.default_rng(seed)
    parts = [np.zeros(int(rng.integers(0, 700)), complex)]
    for i in range(int(rng.integers(3, 40))):
        pay = synth.splitmix64_bytes(seed * 1000 + i, 1, int(rng.integers(1, 400)))[0]
        f = synth.build_frames(pay[None, :], int(rng.choice((0, 2, 3, 5, 6, 8, 9, 10))))[0]
        f = f * np.exp(1j * rng.uniform(0, 6.28)) * 10 ** rng.uniform(-0.7, 0.7)
        if rng.random() < 0.5:
            f = f * np.exp(2j * np.pi * rng.uniform(-4000, 4000) * np.arange(f.size) / 20e6)
        if rng.random() < 0.08 and f.size > 800:
            f = f[:int(rng.integers(400, f.size - 100))]
        snr = rng.uniform(6.0, 30.0)
        sigma = np.sqrt(np.mean(np.abs(f[:320]) ** 2) / (2 * 10 ** (snr / 10)))
        gap = 0 if rng.random() < 0.3 else int(rng.integers(1, 1500))
        seg = np.concatenate([f, np.zeros(gap, complex)])
        parts.append(seg + (rng.normal(size=seg.size) + 1j * rng.normal(size=seg.size)) * sigma)
    parts.append(np.zeros(int(rng.integers(700, 1500)), complex))
    s = np.concatenate(parts).astype(np.complex64)
    want = po.ReceiverChain().run_stream(s.astype(np.complex128))
    src = os.path.join(tmp, "cap.fc32")
    s.tofile(src)
    for mode, extra in (("sync", []), ("async", ["--async", str(int(rng.choice((2, 4, 8))))]), ("device", ["--device-batch", str(int(rng.choice((8192, 32768))))])):
        chunk = int(rng.choice((4096, 4096, 1000, 2500, 7777)))
        out = os.path.join(tmp, "psdus.bin")
        r = subprocess.run([exe, src, "--format", "fc32", "--out", out, "--chunk", str(chunk)] + extra, capture_output=True, text=True, timeout=300)
        runs += 1
        got = []
        if r.returncode == 0:
            raw, o = open(out, "rb").read(), 0
            while o < len(raw):
                n = int.from_bytes(raw[o:o + 4], "little")
                got.append(raw[o + 4:o + 4 + n])
                o += 4 + n
        if r.returncode != 0 or got != want:
            bad += 1
            print("FAIL seed", seed, mode, "chunk", chunk, "rc", r.returncode, "oracle", len(want), "payloads, got", len(got), r.stderr[-200:])
print("seeds %d..%d done: %d runs; runs whose payload list differs from the oracle chain's: %d" % (lo, hi - 1, runs, bad))
