#!/usr/bin/env python3
"""Per-rate kernel times of a decode call (HIP events on the kernels' own stream): every kernel with the machine to itself (calls in
line) and under the pipelined loop, for the eight standard rates at one payload length.  What bench.py's per-rate `roofline` objects are
built from; prints one JSON line per rate.

    python tools/rate_kernels.py [--frames N] [--length L] [--rates 0,2,...] [--max-dbps-hint]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=10240)
    ap.add_argument("--length", type=int, default=4092)
    ap.add_argument("--rates", default="0,2,3,5,6,8,9,10")
    ap.add_argument("--budget", type=int, default=300 * 1000 * 1000, help="cap on samples per call (0 = none)")
    ap.add_argument("--hint", action="store_true", help="set option max_dbps to the rate's dbps (work sets sized for the call's own rate)")
    ap.add_argument("--reps", type=int, default=24)
    ap.add_argument("--alone-only", action="store_true", help="calls in line only (for counter runs)")
    ap.add_argument("--depth", type=int, default=0, help="library option depth for the pipelined part (0: by grid size)")
    args = ap.parse_args()
    import torch
    import fun_ofdm_amd as foa
    from fun_ofdm_amd import synth
    dev = torch.device("cuda", 0)
    DBPS = (24, 32, 36, 48, 64, 72, 96, 128, 144, 192, 216)
    for rate in [int(x) for x in args.rates.split(",")]:
        rx = foa.Receiver(0)
        nsym = -(-(16 + 8 * (args.length + 4) + 6) // DBPS[rate])
        s0 = 320 + 80 * (1 + nsym)
        pitch = -(-(s0 + 576) // 4096) * 4096
        n = args.frames if not args.budget else int(min(args.frames, args.budget // pitch))
        pays = synth.splitmix64_bytes(0x0FD3 + rate, n, args.length)
        frames = rx.tx_build_frames(torch.from_numpy(pays).to(dev), rate)
        s = frames.shape[1]
        assert s == s0, (s, s0)
        d_iq = rx.tx_channel(frames, pitch, 176, 25.0, seed=400 + rate)
        del frames
        cap = d_iq.shape[0] // 512 + 64
        d_desc = torch.zeros(cap * 48, dtype=torch.uint8, device=dev)
        d_end = torch.zeros(cap, dtype=torch.int64, device=dev)
        m = rx.sync_dev(d_iq, d_desc, d_end)
        d_psdu = torch.zeros((m, args.length), dtype=torch.uint8, device=dev)
        d_res = torch.zeros((m, 4), dtype=torch.int32, device=dev)
        if args.hint:
            rx.set_option("max_dbps", DBPS[rate])

        def call():
            rx.decode_frames_dev(d_iq, d_desc[:m * 48], d_end[:m], d_psdu, d_res)

        row = {"rate": rate, "mbps": foa.RATE_MBPS[rate], "frames": n, "alignments": m, "frame_samples": int(s), "symbols": n * nsym, "steps": n * nsym * DBPS[rate],
               "forward_waves_per_simd": round(n / 2 / 1024, 2)}
        rx.set_option("pipeline", 0)
        acc = {}
        for j in range(4):
            call(); rx.sync()
            if j:
                for k, v in rx.kernel_ms().items():
                    acc[k] = acc.get(k, 0.0) + v / 3.0
        row["alone_ms"] = {k: round(v, 4) for k, v in acc.items()}
        if args.alone_only:
            print(json.dumps(row), flush=True)
            rx.close()
            continue
        rx.set_option("pipeline", 1)
        if args.depth:
            rx.set_option("depth", args.depth)
        for _ in range(8):
            call()
        rx.sync(); torch.cuda.synchronize()
        rounds, kacc, kn = [], {}, 0
        for _ in range(3):
            t0 = time.perf_counter()
            for i in range(args.reps):
                call()
                if i >= 3:
                    for k, v in rx.kernel_ms(age=2).items():
                        kacc[k] = kacc.get(k, 0.0) + v
                    kn += 1
            rx.sync(); torch.cuda.synchronize()
            rounds.append((time.perf_counter() - t0) / args.reps)
        dt = sorted(rounds)[1]
        row["piped_ms_per_call"] = round(dt * 1e3, 4)
        row["piped_kernel_ms"] = {k: round(v / kn, 4) for k, v in kacc.items()}
        row["Msamples_per_s"] = round(n * s / dt / 1e6, 1)
        r = d_res.cpu().numpy()
        row["ok"] = int((r[:, 0] == 0).sum())
        row["no_space"] = int((r[:, 0] == 4).sum())
        free, total = torch.cuda.mem_get_info()
        row["hbm_used_GB"] = round((total - free) / 1e9, 2)
        print(json.dumps(row), flush=True)
        rx.close()
        del d_iq, d_desc, d_end, d_psdu, d_res
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
