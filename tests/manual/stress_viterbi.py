"""Random soft-byte blocks through foa_conv_decode (the production forward pass + segment chain-back, random segment / run-in
settings) against the oracle's decoder -- and against the compiled reference decoder when oracle/_ref is there: lengths from 1 to
32 900 data bits (the longest PSDU), 1 .. 9 blocks per call (odd counts leave a wave with one frame), soft bytes drawn from a different distribution per
block: uniform, pushed to the extremes, noisy codewords at random noise levels, erasure bursts, constant runs, saw-tooth ramps --
the inputs on which uint8 saturation, the state-0 renormalisation rule and the tie rule decide the output.
Usage (GPU box, from the repo root): python3 tests/manual/stress_viterbi.py [first seed] [last seed]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fun_ofdm_amd as foa
from oracle import pyoracle as po


def run(lo, hi):
    rx = foa.Receiver(0)
    real = po.Ref.conv_decode if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(po.__file__)), "_ref", "libfun_ofdm_ref.so")) else None
    bad = 0
    nblocks = 0
    for seed in range(lo, hi):
        rng = np.random.default_rng(seed)
        nb = int(rng.choice((rng.integers(1, 200), rng.integers(200, 3000), rng.integers(3000, 32900))))
        n = 2 * (nb + 6)
        blocks = []
        for b in range(int(rng.integers(1, 10))):
            kind = int(rng.integers(0, 7))
            if kind == 0:
                blk = rng.integers(0, 256, n)
            elif kind == 1:
                blk = np.where(rng.random(n) < 0.5, rng.integers(0, 12, n), rng.integers(244, 256, n))
            elif kind in (2, 3):
                d = rng.integers(0, 256, (nb + 13) // 8 + 1, dtype=np.uint8)
                blk = np.clip(po.conv_encode(d, nb).astype(float) * 255 + rng.normal(0, rng.uniform(0, 160), n), 0, 255)
                if kind == 3:                                 # erasure bursts
                    for _ in range(int(rng.integers(1, 8))):
                        a = int(rng.integers(0, n)); blk[a:a + int(rng.integers(1, 400))] = 127
            elif kind == 4:
                blk = np.repeat(rng.integers(0, 256, n // 37 + 1), 37)[:n]
            elif kind == 5:
                blk = (np.arange(n) * int(rng.integers(1, 255))) % 256
            else:
                blk = np.full(n, int(rng.choice((0, 1, 126, 127, 128, 129, 254, 255))))
            blocks.append(np.asarray(blk).astype(np.uint8))
        S = int(rng.choice((96, 192, 960, 3072)))
        L = int(rng.choice((0, 96, 192)))
        rx.set_option("tb_segment", S); rx.set_option("tb_overlap", L)
        got = rx.conv_decode(np.concatenate(blocks), nb, len(blocks))
        for b, blk in enumerate(blocks):
            nblocks += 1
            want = po.conv_decode(blk, nb)
            ok = np.array_equal(got[b], want) and (real is None or np.array_equal(want, real(blk, nb)))
            if not ok:
                bad += 1
                print("FAIL seed", seed, "data bits", nb, "block", b, "segment", S, "run-in", L)
    rx.close()
    return nblocks, bad, real is not None


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lo = int(args[0]) if len(args) > 0 else 0
    hi = int(args[1]) if len(args) > 1 else 100
    nblocks, bad, with_ref = run(lo, hi)
    print("seeds %d..%d done: %d blocks; blocks that differ from the oracle%s: %d" % (lo, hi - 1, nblocks, " or the compiled reference" if with_ref else "", bad))
