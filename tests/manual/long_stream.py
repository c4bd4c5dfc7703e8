"""One stream of more than 2^32 samples through the stream engine (positions beyond 32 bits): a 29.5 M-sample capture of 3600 frames pushed\nagain and again; every repetition must hand out the same 3600 payloads.  Usage (GPU box, repo root): python3 tests/manual/long_stream.py [repetitions, default 160 = 4.7 G samples]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import fun_ofdm_amd as foa
from fun_ofdm_amd import synth
pays = synth.splitmix64_bytes(5, 600, 200)
iq, _ = synth.make_stream(synth.build_frames(pays, 8), 8192, 300, 25.0, seed=2)      # 4.9 M samples, 600 frames
iq = np.concatenate([iq] * 6).astype(np.complex64)                                     # 29.5 M samples, 3600 frames
want = [p.tobytes() for p in pays] * 6
rx = foa.Receiver(0)
st = foa.Stream(rx, 4 << 20, 4)
t0 = time.time(); total = 0; got_all = 0; bad = 0; pending = []
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 160
for r in range(reps):
    pending += st.push(iq)
    total += iq.size
    while len(pending) >= len(want):
        chunk, pending = pending[:len(want)], pending[len(want):]
        got_all += len(chunk)
        if chunk != want:
            bad += 1; print("MISMATCH in repetition ending near sample", total, "first diff", next(i for i in range(len(want)) if chunk[i] != want[i]))
pending += st.flush()
print("pushed %.3f G samples (2^31 = 2.147 G, 2^32 = 4.295 G) in %.1f s; payloads %d + %d pending of %d expected; repetitions with a wrong list: %d"
      % (total / 1e9, time.time() - t0, got_all, len(pending), reps * len(want), bad))
print("stats", st.stats() if hasattr(st, "stats") else "")
