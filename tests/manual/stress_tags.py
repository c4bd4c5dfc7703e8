"""Made-up alignments: what fft_symbols .. frame_decoder do with a GIVEN tag stream (oracle blocks, po.chain_from_tags_f32) against the batch
restatement of it (po.decode_batch_v2_f32) and, in gpu mode, against the device's batch path handed the same descriptors.

The physical streams of stress_collide.py hardly ever put an LTS1 tag where the partial-vector flush of fft_symbols.cpp:41-50 decides a
frame's fate (the detector needs the second preamble ~10 dB above the first frame, which ruins the symbols under it).  Here the tags are
PLACED: on a clean multi-frame stream, extra alignments are inserted at chosen sample offsets -- late in a frame's last symbol (the partial
vector completes the frame), one symbol earlier (the partial vector and the next alignment's SIGNAL complete it), anywhere (the frame is
dropped or fails), on noise, and in PILE-UPS (one to three more tags within 1 .. 207 samples of an alignment, a true one included: cut
LTS windows, a SIGNAL window that is only part fresh, an LTS2 tag inside the next alignment's first LTS window -- which moves that
alignment's vectors one symbol later) -- with phasors chained as timing_sync would chain them.
Usage: python3 tests/manual/stress_tags.py cpu|gpu [first seed] [last seed]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import pyoracle as po

RATES = (0, 2, 3, 5, 6, 8, 9, 10)


def make_case(seed):
    rng = np.random.default_rng(seed)
    parts, pos = [np.zeros(int(rng.integers(300, 700)), complex)], []
    at = parts[0].size
    for i in range(int(rng.integers(2, 6))):
        rate = int(rng.choice(RATES))
        length = int(rng.integers(1, 300))
        f = po.build_frame(rng.integers(0, 256, length, dtype=np.uint8), rate) * np.exp(1j * rng.uniform(0, 6.28))
        gap = int(rng.integers(0, 500))
        parts += [f, np.zeros(gap, complex)]
        pos.append((at, f.size))
        at += f.size + gap
    parts.append(np.zeros(600, complex))
    s = np.concatenate(parts)
    s = (s + (rng.normal(size=s.size) + 1j * rng.normal(size=s.size)) * np.sqrt(0.0124 / 2 / 10 ** rng.uniform(1.8, 3.2))).astype(np.complex64)
    # the true alignments (timing_sync tags LTS1 184 samples into a preamble), then the made-up ones
    lts = [(a + 184, True) for a, _ in pos]
    for a, size in pos:
        end = a + size                                   # the frame's last symbol window starts 72 before its end
        kind = rng.choice(("late", "prev", "any", "none"), p=(0.4, 0.2, 0.3, 0.1))
        if kind == "late":
            lts.append((end - 72 + int(rng.integers(16, 72)), False))
        elif kind == "prev":
            lts.append((end - 152 + int(rng.integers(0, 80)), False))
        elif kind == "any":
            lts.append((int(rng.integers(a + 400, max(a + 401, end + 100))), False))
    if rng.random() < 0.3:
        lts.append((int(rng.integers(0, s.size - 300)), False))
    for p, real in list(lts):                            # pile-ups: more tags close to one that is there
        if rng.random() < 0.12:
            for _ in range(int(rng.integers(1, 4))):
                lts.append((p + int(rng.choice((-1, 1))) * int(rng.choice((rng.integers(1, 64), rng.integers(64, 128), rng.integers(128, 208)))), False))
    lts.sort()
    keep = []
    for p, real in lts:                                  # one tag per sample (a true alignment wins)
        if keep and p == keep[-1][0]:
            keep[-1] = (p, real or keep[-1][1])
            continue
        if 0 <= p < s.size - 130:
            keep.append((p, real))
    d = np.zeros(len(keep), po.frame_desc)
    c_prev, s_prev = 1.0, 0.0
    for j, (p, real) in enumerate(keep):
        ph = rng.uniform(-3.1, 3.1)
        d[j]["lts1_pos"] = p
        d[j]["rot_start"] = p + int(rng.integers(0, 9))
        d[j]["c"], d[j]["s"] = np.cos(ph), np.sin(ph)
        d[j]["c_prev"], d[j]["s_prev"] = c_prev, s_prev
        c_prev, s_prev = d[j]["c"], d[j]["s"]
    return s, d


def payloads(psdu, res):
    return [psdu[f, :res[f]["length"]].tobytes() for f in range(res.size) if res[f]["status"] == 0]


def alone_misses(s, d, ends, res):
    """frames only the flush-aware restatement delivers: CRC-passing there, not decodable from their own alignment's samples alone"""
    n = 0
    for j in np.nonzero(res["status"] == 0)[0]:
        r1, _ = po.decode_alignment_f32(s, d[j], end=int(ends[j]))
        n += int(r1["status"] != 0)
    return n


def run_cpu(lo, hi):
    bad = tot = n_al = flush_hits = 0
    for seed in range(lo, hi):
        s, d = make_case(seed)
        want = po.chain_from_tags_f32(s, d)
        psdu, res = po.decode_batch_v2_f32(s, d)
        got = payloads(psdu, res)
        ends = np.append(d["lts1_pos"][1:], s.size).astype(np.int64)
        psdu2, res2 = po.decode_batch_f32(s, d, ends)            # every alignment on its own + the fix-up pass: must be the same thing
        if not (np.array_equal(res.view(np.int32), res2.view(np.int32)) and np.array_equal(psdu, psdu2)):
            bad += 1
            print("DIFF seed", seed, "decode_batch_f32 and decode_batch_v2_f32 disagree", res["status"].tolist(), res2["status"].tolist())
        flush_hits += alone_misses(s, d, ends, res)
        tot += len(want)
        n_al += d.size
        if got != want:
            bad += 1
            print("DIFF seed", seed, "alignments", d.size, "blocks", len(want), "payloads, restatement", len(got), res["status"].tolist())
    return tot, n_al, bad, flush_hits


def run_gpu(lo, hi):
    import fun_ofdm_amd as foa
    rx = foa.Receiver(0)
    bad = tot = n_al = flush_hits = 0
    for seed in range(lo, hi):
        s, d = make_case(seed)
        want = po.chain_from_tags_f32(s, d)
        opsdu, ores = po.decode_batch_v2_f32(s, d)
        ends = np.append(d["lts1_pos"][1:], s.size).astype(np.int64)
        psdu, res = rx.decode_frames_host(s, d, ends)
        flush_hits += alone_misses(s, d, ends, ores)
        tot += len(want)
        n_al += d.size
        ok = np.array_equal(res.view(np.int32), ores.view(np.int32)) and payloads(psdu, res) == want
        if not ok:
            bad += 1
            print("DIFF seed", seed, "blocks", len(want), "device", len(payloads(psdu, res)), "device statuses", res["status"].tolist(), "restatement", ores["status"].tolist())
    rx.close()
    return tot, n_al, bad, flush_hits


def rotated(s, d):
    """the stream as timing_sync hands it on for these alignments: every sample times the phasor in force at its index (timing_sync.cpp:121-125)"""
    out = s.astype(np.complex128)
    rot = np.full(s.size, complex(d[0]["c_prev"], d[0]["s_prev"]) if d.size else 1.0 + 0j)
    for j in range(d.size):
        rot[max(int(d[j]["rot_start"]), 0):] = complex(d[j]["c"], d[j]["s"])
    return out * rot


def run_gpu_f64(lo, hi):
    """the same cases as complex<double> samples rotated already (what the fused stage block fun_amd::rx_backend hands the device)"""
    import fun_ofdm_amd as foa
    rx = foa.Receiver(0)
    bad = tot = n_al = 0
    for seed in range(lo, hi):
        s, d = make_case(seed)
        r = rotated(s, d)
        want = po.chain_from_tags_f32(s, d)
        ends = np.append(d["lts1_pos"][1:], s.size).astype(np.int64)
        opsdu, ores = po.decode_batch_v2_f64(r, d, ends)
        psdu, res = rx.decode_frames_f64_host(r, d, ends)
        tot += len(want)
        n_al += d.size
        if not (np.array_equal(res.view(np.int32), ores.view(np.int32)) and np.array_equal(psdu, opsdu) and payloads(psdu, res) == want):
            bad += 1
            print("DIFF (f64) seed", seed, "blocks", len(want), "device", res["status"].tolist(), "restatement", ores["status"].tolist())
    rx.close()
    return tot, n_al, bad


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "cpu"
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    tot, n_al, bad, hits = (run_cpu if mode == "cpu" else run_gpu)(lo, hi)
    print("seeds %d..%d: %d alignments, %d payloads from the blocks, %d of them delivered only because of the partial-vector flush / frame-in-progress rules; cases that differ: %d"
          % (lo, hi - 1, n_al, tot, hits, bad))
