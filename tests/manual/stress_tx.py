"""Random (rate, length, payload) through foa_tx_build_frames_dev against the oracle's build_frame: sample count, preamble identical,
every sample within 1e-12 relative.  Usage (GPU box, from the repo root): python3 tests/manual/stress_tx.py [first seed] [last seed]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import fun_ofdm_amd as foa
from oracle import pyoracle as po


def run(lo, hi):
    rx = foa.Receiver(0)
    bad = 0
    n_frames = 0
    worst = 0.0
    for seed in range(lo, hi):
        rng = np.random.default_rng(seed)
        rate = int(rng.integers(0, 11))
        length = int(rng.choice((0, 1, 2, 4095, 4094, int(rng.integers(3, 300)), int(rng.integers(300, 4094)))))
        k = int(rng.integers(1, 5))
        kind = int(rng.integers(0, 4))
        pays = (rng.integers(0, 256, (k, length)) if kind == 0 else np.full((k, length), int(rng.choice((0, 255, 0x55, 0xAA)))) if kind == 1
                else rng.integers(0, 2, (k, length)) * 255 if kind == 2 else np.tile(np.arange(length) % 256, (k, 1))).astype(np.uint8)
        t = torch.zeros((k, length), dtype=torch.uint8, device="cuda:0")
        t.copy_(torch.from_numpy(np.ascontiguousarray(pays)))
        got = rx.tx_build_frames(t, rate).cpu().numpy()
        got = got[..., 0] + 1j * got[..., 1]
        for i in range(k):
            n_frames += 1
            want = po.build_frame(pays[i], rate)
            ok = got[i].size == want.size == rx.tx_frame_samples(length, rate) and np.array_equal(got[i][:320], want[:320])
            err = np.abs(got[i] - want).max() / np.abs(want).max() if ok else 1.0
            worst = max(worst, err)
            if not ok or err >= 1e-12:
                bad += 1
                print("FAIL seed", seed, "rate", rate, "length", length, "frame", i, "err", err)
    rx.close()
    return n_frames, bad, worst


if __name__ == "__main__":
    lo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    hi = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    n_frames, bad, worst = run(lo, hi)
    print("seeds %d..%d done: %d frames, largest relative sample error %.2e; frames that differ from the oracle: %d" % (lo, hi - 1, n_frames, worst, bad))
