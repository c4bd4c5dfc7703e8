#!/usr/bin/env python3
"""Soak test of the pipelined path (run on the GPU box): a few hundred back-to-back decode calls over batches of different
sizes and rates, device pre-sync in between, no host synchronisation except to read results three calls late; every
call's PSDUs and results must equal those of the same batch decoded alone with the pipeline off."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

import fun_ofdm_amd as foa
from fun_ofdm_amd import synth

dev = torch.device("cuda", 0)
rx = foa.Receiver(0)
rx.set_option("record_soft", 0)
rng = np.random.default_rng(2024)
batches = []
for b in range(6):
    n = int(rng.integers(50, 3000))
    rate = int(rng.choice([0, 2, 3, 5, 6, 8, 9, 10]))
    length = int(rng.integers(20, 1200))
    pays = synth.splitmix64_bytes(100 + b, n, length)
    frames = rx.tx_build_frames(torch.from_numpy(pays).to(dev), rate)
    s = frames.shape[1]
    pitch = -(-(s + 400) // 1024) * 1024
    iq = rx.tx_channel(frames, pitch, 176, 25.0, seed=77 + b)
    cap = n * pitch // 400 + 64
    descs = torch.zeros(cap * 48, dtype=torch.uint8, device=dev)
    ends = torch.zeros(cap, dtype=torch.int64, device=dev)
    rx.set_option("pipeline", 0)
    m = rx.sync_dev(iq, descs, ends)
    psdu = torch.zeros((m, length), dtype=torch.uint8, device=dev)
    res = torch.zeros((m, 4), dtype=torch.int32, device=dev)
    rx.decode_frames_dev(iq, descs[:m * 48], ends[:m], psdu, res)
    rx.sync()
    ok = int((res[:, 0] == 0).sum().item())
    batches.append((iq, descs, ends, m, length, psdu.clone(), res.clone(), descs.clone()))
    print("batch %d: %d frames rate %d length %d -> %d alignments, %d ok" % (b, n, rate, length, m, ok), flush=True)
torch.cuda.synchronize()
rx.set_option("pipeline", 1)
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ring, bad = [], 0


def same(p, r, wp, wr):
    ok = wr[:, 0] == 0                               # slots of frames that failed are not written
    return bool(torch.equal(r, wr) and torch.equal(p[ok], wp[ok]))


# output / descriptor sets are made ahead of the loop (six per batch, reused round robin, refilled by torch right after their
# check): no torch synchronisation inside the loop, which on this runtime would drain the library's streams as well
pools = []
for iq, descs, ends, m, length, want_psdu, want_res, want_descs in batches:
    pools.append([(torch.zeros_like(descs), torch.zeros_like(ends), torch.full((m, length), 0xEE, dtype=torch.uint8, device=dev),
                   torch.full((m, 4), -1, dtype=torch.int32, device=dev)) for _ in range(6)])
uses = [0] * len(batches)
torch.cuda.synchronize()
for c in range(calls):
    b = int(rng.integers(0, len(batches)))
    iq, descs, ends, m, length, want_psdu, want_res, want_descs = batches[b]
    d2, e2, psdu, res = pools[b][uses[b] % 6]
    uses[b] += 1
    if c % 3 == 0:                                   # pre-sync on the device in between (third stream)
        m2 = rx.sync_dev(iq, d2, e2)
        assert m2 == m
    else:
        d2.copy_(want_descs); e2.copy_(ends)
        ev = torch.cuda.Event(); ev.record(); ev.synchronize()    # the copies are through (one event, not a stream synchronize)
    rx.decode_frames_dev(iq, d2[:m * 48], e2[:m], psdu, res)
    ring.append((psdu, res, want_psdu, want_res, d2, e2))
    if len(ring) > 3:
        p, r, wp, wr, _, _ = ring.pop(0)
        rx.wait_age(2)                              # call c-2 complete, hence c-3 too
        if not same(p, r, wp, wr):
            bad += 1
            print("MISMATCH in call", c - 3, flush=True)
        p.fill_(0xEE); r.fill_(-1)
rx.sync()
for p, r, wp, wr, _, _ in ring:
    if not same(p, r, wp, wr):
        bad += 1
print("soak done: %d calls, %d mismatches" % (calls, bad))
sys.exit(1 if bad else 0)
