"""Random streams through process_samples() on the device (the stream engine, foa.Stream: random push sizes, random batch sizes)
against the oracle's receiver_chain fed the way the reference's receiver feeds it (4096 samples per call): the ORDERED PAYLOAD LISTS
must be equal.  Frames of random rate / length / amplitude / phase / CFO, gaps from none to 1500 samples, SNR 6 .. 30 dB (some fail
their CRC on both sides), now and then a frame cut short by the next one's preamble.
Usage (GPU box, from the repo root): python3 tests/manual/stress_stream.py [first seed] [last seed]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fun_ofdm_amd as foa
from fun_ofdm_amd import synth
from oracle import pyoracle as po


def run(lo, hi):
    rx = foa.Receiver(0)
    bad = 0
    tot = 0
    for seed in range(lo, hi):
        rng = np.random.default_rng(seed)
        parts = [np.zeros(int(rng.integers(0, 700)), complex)]
        for i in range(int(rng.integers(1, 30))):
            pay = synth.splitmix64_bytes(seed * 1000 + i, 1, int(rng.integers(1, 400)))[0]
            f = synth.build_frames(pay[None, :], int(rng.choice((0, 2, 3, 5, 6, 8, 9, 10))))[0]
            f = f * np.exp(1j * rng.uniform(0, 6.28)) * 10 ** rng.uniform(-0.7, 0.7)
            if rng.random() < 0.5:
                f = f * np.exp(2j * np.pi * rng.uniform(-4000, 4000) * np.arange(f.size) / 20e6)
            if rng.random() < 0.08 and f.size > 800:
                f = f[:int(rng.integers(400, f.size - 100))]             # cut short: the next preamble arrives inside this frame
            snr = rng.uniform(6.0, 30.0)
            sigma = np.sqrt(np.mean(np.abs(f[:320]) ** 2) / (2 * 10 ** (snr / 10)))
            gap = 0 if rng.random() < 0.3 else int(rng.integers(1, 1500))
            seg = np.concatenate([f, np.zeros(gap, complex)])
            parts.append(seg + (rng.normal(size=seg.size) + 1j * rng.normal(size=seg.size)) * sigma)
        parts.append(np.zeros(int(rng.integers(200, 900)), complex))
        s = np.concatenate(parts).astype(np.complex64)
        want = po.ReceiverChain().run_stream(s.astype(np.complex128))
        st = foa.Stream(rx, int(rng.choice((4096, 8192, 16384, 65536))), int(rng.integers(0, 3)))
        got, i = [], 0
        while i < s.size:
            n = int(rng.integers(1, 20000))
            got += st.push(s[i:i + n])
            i += n
        got += st.flush()
        st.close()
        tot += len(want)
        if got != want:
            bad += 1
            print("FAIL seed", seed, "samples", s.size, "oracle", len(want), "payloads, stream engine", len(got),
                  "first difference at", next((k for k in range(min(len(want), len(got))) if want[k] != got[k]), min(len(want), len(got))))
    rx.close()
    return tot, bad


if __name__ == "__main__":
    lo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    hi = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    tot, bad = run(lo, hi)
    print("seeds %d..%d done: %d payloads in all; streams whose payload list differs: %d" % (lo, hi - 1, tot, bad))
