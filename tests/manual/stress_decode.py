"""Perturbed and made-up alignment descriptors through foa_rx_decode_frames_host against the oracle's per-alignment decoder on the
SAME descriptors: true alignments shifted by a few samples (so that SIGNAL fields decode to other rates and lengths, or not at all),
phasors off the unit circle, ends cut anywhere (truncated frames), positions in noise, in the middle of payloads, at the very start
and end of the stream; SNR from 4 dB up.  Status, rate, length, symbol count and every PSDU that passes must be identical.
Usage (GPU box, from the repo root): python3 tests/manual/stress_decode.py [first seed] [last seed] [amplitude exponent range, default 0.5]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fun_ofdm_amd as foa
from fun_ofdm_amd import synth
from oracle import pyoracle as po


def run(lo, hi, amp=0.5):
    rx = foa.Receiver(0)
    bad = 0
    tot = 0
    passed = 0
    for seed in range(lo, hi):
        rng = np.random.default_rng(seed)
        parts = [np.zeros(int(rng.integers(0, 500)), complex)]
        for i in range(int(rng.integers(2, 25))):
            pay = synth.splitmix64_bytes(seed * 1000 + i, 1, int(rng.choice((0, 1, int(rng.integers(2, 300)), int(rng.integers(300, 1500))))))[0]
            f = synth.build_frames(pay[None, :], int(rng.integers(0, 11)))[0] * np.exp(1j * rng.uniform(0, 6.28)) * 10 ** rng.uniform(-amp, amp)
            if rng.random() < 0.5:
                f = f * np.exp(2j * np.pi * rng.uniform(-4500, 4500) * np.arange(f.size) / 20e6)
            snr = rng.uniform(4.0, 30.0)
            sigma = np.sqrt(np.mean(np.abs(f[:320]) ** 2) / (2 * 10 ** (snr / 10)))
            seg = np.concatenate([f, np.zeros(int(rng.integers(0, 800)), complex)])
            parts.append(seg + (rng.normal(size=seg.size) + 1j * rng.normal(size=seg.size)) * sigma)
        s = np.concatenate(parts).astype(np.complex64)
        n = s.size
        true = po.find_alignments_f32(s)
        d = []
        for a in true:
            for _ in range(int(rng.integers(1, 4))):
                b = a.copy()
                sh = int(rng.choice((0, 0, 1, -1, 2, -2, 3, -5, 8, 16, -16, 64, 80)))
                b["lts1_pos"] = min(max(int(a["lts1_pos"]) + sh, 0), n - 1)
                b["rot_start"] = min(max(int(a["rot_start"]) + int(rng.integers(-40, 40)), 0), n - 1)
                if rng.random() < 0.3:
                    ph, g = rng.uniform(0, 6.28), rng.choice((1.0, 1.0, 0.5, 2.0, 1e-3, 1e3))
                    b["c"], b["s"] = g * np.cos(ph), g * np.sin(ph)
                if rng.random() < 0.3:
                    ph = rng.uniform(0, 6.28)
                    b["c_prev"], b["s_prev"] = np.cos(ph), np.sin(ph)
                d.append(b)
        for _ in range(int(rng.integers(0, 12))):             # anywhere at all
            b = np.zeros(1, foa.frame_desc_dtype)[0]
            b["lts1_pos"] = int(rng.integers(0, n)); b["rot_start"] = int(rng.integers(0, n))
            b["c"], b["s"], b["c_prev"], b["s_prev"] = 1.0, 0.0, 1.0, 0.0
            d.append(b)
        if not d:
            continue
        d = np.array(d, foa.frame_desc_dtype)
        d = d[np.argsort(d["lts1_pos"], kind="stable")]
        ends = np.empty(d.size, np.int64)
        for k in range(d.size):
            nxt = int(d["lts1_pos"][k + 1]) if k + 1 < d.size else n
            lo_e = int(d["lts1_pos"][k]) + 1
            ends[k] = nxt if rng.random() < 0.7 else int(rng.integers(lo_e, max(lo_e + 1, min(n, lo_e + 6000)) + 0))
            ends[k] = min(max(ends[k], lo_e), n)
        opsdu, ores = po.decode_batch_f32(s, d, ends, threads=8)
        psdu, res = rx.decode_frames_host(s, d, ends)
        tot += d.size
        same = np.array_equal(res.view(np.int32), ores.view(np.int32))
        ok = res["status"] == 0
        same = same and np.array_equal(psdu[ok], opsdu[ok])
        passed += int(ok.sum())
        if not same:
            bad += 1
            w = np.nonzero((res.view(np.int32).reshape(-1, 4) != ores.view(np.int32).reshape(-1, 4)).any(axis=1))[0]
            print("FAIL seed", seed, "alignments", d.size, "first differing", w[:3], res[w[:3]] if w.size else "", ores[w[:3]] if w.size else "(PSDU bytes)")
    rx.close()
    return tot, passed, bad


if __name__ == "__main__":
    lo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    hi = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    amp = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
    tot, passed, bad = run(lo, hi, amp)
    print("seeds %d..%d done: %d descriptors, %d of them pass their CRC; streams with any difference from the oracle: %d" % (lo, hi - 1, tot, passed, bad))
