"""Dev container only (needs /root/reference compiled into oracle/_ref): random streams through the COMPILED reference's
frame_detector, timing_sync, channel_est and phase_tracker against the oracle's restatements of them, call by call, tag for tag and
sample for sample (4096 samples per call like the reference's receiver; fft_symbols between them is the oracle's on both sides -- the
reference's needs FFTW); and random blocks through the compiled codec pieces (puncture / depuncture / interleave / deinterleave / modulate / demodulate /
conv_encode / conv_decode) against the oracle's.  This pins the ORACLE; the HIP path is pinned against the oracle on the GPU box.
Usage: python3 tests/manual/stress_oracle_vs_ref.py [first seed] [last seed]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import pyoracle as po


def random_stream(rng):
    parts = [np.zeros(int(rng.integers(0, 700)), complex)]
    for i in range(int(rng.integers(1, 25))):
        f = po.build_frame(rng.integers(0, 256, int(rng.integers(1, 500)), dtype=np.uint8), int(rng.integers(0, 11)))
        f = f * np.exp(1j * rng.uniform(0, 6.28)) * 10 ** rng.uniform(-1.0, 1.0)
        if rng.random() < 0.5:
            f = f * np.exp(2j * np.pi * rng.uniform(-5000, 5000) * np.arange(f.size) / 20e6)
        if rng.random() < 0.1 and f.size > 800:
            f = f[:int(rng.integers(400, f.size - 100))]
        snr = rng.uniform(3.0, 30.0)
        sigma = np.sqrt(np.mean(np.abs(f[:320]) ** 2) / (2 * 10 ** (snr / 10)))
        seg = np.concatenate([f, np.zeros(0 if rng.random() < 0.3 else int(rng.integers(1, 1500)), complex)])
        parts.append(seg + (rng.normal(size=seg.size) + 1j * rng.normal(size=seg.size)) * sigma)
    s = np.concatenate(parts).astype(np.complex64).astype(np.complex128)
    if rng.random() < 0.15:
        s[int(rng.integers(0, s.size))] = np.nan
    if rng.random() < 0.05:
        s[int(rng.integers(0, s.size))] = 1e10
    n = (s.size // 4096 + 2) * 4096
    return np.concatenate([s, np.zeros(n - s.size)])


def run(lo, hi):
    R = po.Ref
    bad = calls = 0
    for seed in range(lo, hi):
        rng = np.random.default_rng(seed)
        s = random_stream(rng)
        fd_o, ts_o, fs, ce_o, pt_o = po.FrameDetector(), po.TimingSync(), po.FFTSymbols(), po.ChannelEst(), po.PhaseTracker()
        fd_r, ts_r, ce_r, pt_r = (R.Block(k) for k in ("frame_detector", "timing_sync", "channel_est", "phase_tracker"))
        ok = True
        for x in range(0, s.size, 4096):
            calls += 1
            a, b = fd_o.work(s[x:x + 4096]), fd_r.work(s[x:x + 4096])
            ok = ok and np.array_equal(a["tag"], b["tag"]) and np.array_equal(a["sample"], b["sample"], equal_nan=True)
            c, d = ts_o.work(a), ts_r.work(b)
            ok = ok and np.array_equal(c["tag"], d["tag"]) and np.array_equal(c["sample"], d["sample"], equal_nan=True)
            v = fs.work(c)
            e, f = ce_o.work(v), ce_r.work(v)
            ok = ok and np.array_equal(e["tag"], f["tag"]) and np.array_equal(e["samples"], f["samples"], equal_nan=True)
            g, h = pt_o.work(e), pt_r.work(f)
            ok = ok and np.array_equal(g["tag"], h["tag"]) and np.array_equal(g["samples"], h["samples"], equal_nan=True)
            if not ok:
                break
        # codec pieces on random blocks
        rate = int(rng.integers(0, 11))
        nb = int(rng.integers(1, 3000))
        d8 = rng.integers(0, 256, (nb + 13) // 8 + 1, dtype=np.uint8)
        enc = po.conv_encode(d8, nb)
        ok = ok and np.array_equal(enc, R.conv_encode(d8, nb))
        soft = rng.integers(0, 256, 2 * (nb + 6), dtype=np.uint8)
        ok = ok and np.array_equal(po.conv_decode(soft, nb), R.conv_decode(soft, nb))
        car = (rng.normal(size=96) + 1j * rng.normal(size=96)) * 10 ** rng.uniform(-3, 3)
        ok = ok and np.array_equal(po.demodulate(car, rate), R.demodulate(car, rate))
        rp = po.rate_params(rate)
        k = int(rng.integers(1, 9))
        by = rng.integers(0, 256, rp["cbps"] * k, dtype=np.uint8)
        ok = ok and np.array_equal(po.deinterleave(by), R.deinterleave(by)) and np.array_equal(po.interleave(by), R.interleave(by))
        ok = ok and np.array_equal(po.depuncture(by, rate), R.depuncture(by, rate))
        bits = rng.integers(0, 2, rp["dbps"] * 2 * k, dtype=np.uint8)
        ok = ok and np.array_equal(po.puncture(bits, rate), R.puncture(bits, rate))
        mb = rng.integers(0, 2, rp["cbps"] * k, dtype=np.uint8)
        ok = ok and np.array_equal(po.modulate(mb, rate), R.modulate(mb, rate))
        if not ok:
            bad += 1
            print("FAIL seed", seed)
    return calls, bad


if __name__ == "__main__":
    lo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    hi = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    if not po.ref_available():
        sys.exit("the compiled reference (oracle/_ref) is not here")
    calls, bad = run(lo, hi)
    print("seeds %d..%d done: %d calls of 4096 samples through four blocks each; streams on which the oracle differs from the compiled reference: %d" % (lo, hi - 1, calls, bad))
