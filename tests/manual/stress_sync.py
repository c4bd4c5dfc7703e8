"""Random streams through the device pre-sync (foa_rx_sync_dev, blocking and in two halves) against the oracle's frame_detector +
timing_sync fed the way the reference's receiver feeds them, 4096 samples per call (oracle/pyoracle.find_alignments_f32), and against
the library's host restatement (foa_sync_push_*, one push + a flush of zeros).  All three must agree, including on the one line of
timing_sync that depends on where the calls' boundaries fall (`if(lts_offset < 0) break`, timing_sync.cpp:99: an alignment whose
STS_END is the first sample a call walks over is dropped when its tag is a sample late), which the library decides for the
reference's call size whatever the sizes it is fed with (foa_sync_set_call / option "sync_call").  Frames of random rate / length / amplitude / phase / CFO at random
gaps (including none), random SNR between 3 and 30 dB, random stream lengths around the flag kernel's 1024-sample groups, now and
then a NaN sample.  Usage (GPU box, from the repo root): python3 tests/manual/stress_sync.py [first seed] [last seed]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import fun_ofdm_amd as foa
from fun_ofdm_amd import synth
from oracle import pyoracle as po


def run(lo, hi):
    dev = torch.device("cuda", 0)
    rx = foa.Receiver(0)
    bad = 0
    tot = 0
    quirk = 0
    for seed in range(lo, hi):
        rng = np.random.default_rng(seed)
        parts = [np.zeros(int(rng.integers(0, 700)), complex)]
        for i in range(int(rng.integers(1, 40))):
            pay = synth.splitmix64_bytes(seed * 1000 + i, 1, int(rng.integers(1, 600)))[0]
            f = synth.build_frames(pay[None, :], int(rng.choice((0, 2, 3, 5, 6, 8, 9, 10))))[0]
            f = f * np.exp(1j * rng.uniform(0, 6.28)) * 10 ** rng.uniform(-1.0, 1.0)
            if rng.random() < 0.5:
                f = f * np.exp(2j * np.pi * rng.uniform(-5000, 5000) * np.arange(f.size) / 20e6)
            snr = rng.uniform(3.0, 30.0)
            sigma = np.sqrt(np.mean(np.abs(f[:320]) ** 2) / (2 * 10 ** (snr / 10)))
            gap = 0 if rng.random() < 0.3 else int(rng.integers(1, 1500))
            seg = np.concatenate([f, np.zeros(gap, complex)])
            parts.append(seg + (rng.normal(size=seg.size) + 1j * rng.normal(size=seg.size)) * sigma)
        s = np.concatenate(parts)
        cut = int(rng.integers(0, 1100))                     # end anywhere relative to the kernel's groups, also inside a frame
        if cut < s.size - 400:
            s = s[:s.size - cut]
        s = s.astype(np.complex64)
        if rng.random() < 0.2:
            s[int(rng.integers(0, s.size))] = np.nan
        want = po.find_alignments_f32(s)
        host = foa.find_alignments(s)
        if host.size != want.size or not (np.array_equal(host["lts1_pos"], want["lts1_pos"]) and np.array_equal(host["rot_start"], want["rot_start"])):
            quirk += 1
            print("note: seed", seed, "host restatement", host.size, "alignments, oracle", want.size)
        t_iq = torch.from_numpy(s.view(np.float32).reshape(-1, 2)).to(dev)
        cap = s.size // 200 + 64
        ok = True
        for mode in (0, 1):
            d = torch.zeros(cap * 48, dtype=torch.uint8, device=dev)
            e = torch.zeros(cap, dtype=torch.int64, device=dev)
            if mode == 0:
                n = rx.sync_dev(t_iq, d, e)
            else:
                rx.sync_dev_begin(t_iq, d, e)
                n = rx.sync_dev_end()
            got = d.cpu().numpy()[:n * 48].view(foa.frame_desc_dtype)
            same = n == want.size and np.array_equal(got["lts1_pos"], want["lts1_pos"]) and np.array_equal(got["rot_start"], want["rot_start"])
            same = same and (n == 0 or max(np.abs(got[k] - want[k]).max() for k in ("c", "s", "c_prev", "s_prev")) < 1e-12)
            same = same and np.array_equal(e.cpu().numpy()[:n], foa.alignment_ends(want, s.size))
            ok = ok and same
        tot += want.size
        if not ok:
            bad += 1
            print("FAIL seed", seed, "samples", s.size, "alignments", want.size)
    rx.close()
    return tot, bad, quirk


if __name__ == "__main__":
    lo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    hi = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    tot, bad, quirk = run(lo, hi)
    print("seeds %d..%d done: %d alignments in all; streams on which the device differs from the oracle: %d; on which the host restatement does: %d"
          % (lo, hi - 1, tot, bad, quirk))
