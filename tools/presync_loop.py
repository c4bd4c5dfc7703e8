#!/usr/bin/env python3
"""BASELINE config 2 with frame_detector + timing_sync on the device inside the loop, the way bench.py's `incl_device_pre_sync.pipelined` leg runs
it: per step foa_rx_sync_dev_end(k), foa_rx_sync_dev_begin(k+1), foa_rx_decode_frames_dev(k).  Prints ms per step; run under
`rocprofv3 --kernel-trace` to see where the pre-sync kernels land against the forward pass (tools/trace_timeline.py).

    python tools/presync_loop.py [--frames N] [--steps K] [--no-presync]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=10000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--no-presync", action="store_true")
    ap.add_argument("--depth", type=int, default=0)
    ap.add_argument("--decode-first", action="store_true", help="per step: end(k), decode(k), begin(k+1) instead of end(k), begin(k+1), decode(k)")
    args = ap.parse_args()
    import torch
    import fun_ofdm_amd as foa
    from fun_ofdm_amd import synth
    dev = torch.device("cuda", 0)
    rx = foa.Receiver(0)
    pays = synth.splitmix64_bytes(0xF00D, args.frames, 1024)
    frames = rx.tx_build_frames(torch.from_numpy(pays).to(dev), 10)
    d_iq = rx.tx_channel(frames, 4096, 176, 25.0, seed=7919)
    del frames
    cap = d_iq.shape[0] // 300 + 16
    sets = [(torch.zeros(cap * 48, dtype=torch.uint8, device=dev), torch.zeros(cap, dtype=torch.int64, device=dev)) for _ in range(2)]
    m = rx.sync_dev(d_iq, *sets[0])
    rx.sync_dev(d_iq, *sets[1])
    psdu = torch.zeros((m, 1024), dtype=torch.uint8, device=dev)
    res = torch.zeros((m, 4), dtype=torch.int32, device=dev)
    rx.set_option("record_soft", 0)
    if args.depth:
        rx.set_option("depth", args.depth)
    rx.reserve(d_iq.shape[0], m)
    for timed in (False, True):
        rx.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        if not args.no_presync:
            rx.sync_dev_begin(d_iq, *sets[0])
        for k in range(args.steps):
            nf = m
            if not args.no_presync:
                nf = rx.sync_dev_end()
                if k + 1 < args.steps and not args.decode_first:
                    rx.sync_dev_begin(d_iq, *sets[(k + 1) % 2])
            dsc, en = sets[k % 2]
            rx.decode_frames_dev(d_iq, dsc[:nf * 48], en[:nf], psdu[:nf], res[:nf], settle=False)
            if not args.no_presync and k + 1 < args.steps and args.decode_first:
                rx.sync_dev_begin(d_iq, *sets[(k + 1) % 2])
        rx.sync(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
    ok = int((res.cpu().numpy()[:, 0] == 0).sum())
    print("%d frames, %d alignments, %d ok: %.4f ms per step (%s)" % (args.frames, m, ok, dt * 1e3, "decode only" if args.no_presync else "pre-sync + decode"))
    rx.close()


if __name__ == "__main__":
    main()
