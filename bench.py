#!/usr/bin/env python3
"""bench.py -- RX Msamples/s of the MI355X receive hot path on BASELINE.json config 2.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One step = one pass of the hot path (LTS/SIGNAL -> data symbols -> Viterbi/descramble/CRC -> PSDUs)
over one batch of 10 000 synthetic 54 Mbps frames (1024-byte payloads, AWGN 25 dB, one frame per
4096-sample slot), with the sample stream and the alignment descriptors already resident in HBM.
For N > 1 every rank decodes its own 10 000-frame shard (weak scaling; frames are independent, so the
data path has no collective) and the decoded PSDUs are gathered to rank 0 over RCCL inside the step.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RATE, PAYLOAD, SNR_DB, PITCH, LEAD = 10, 1024, 25.0, 4096, 176
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s


def log(*a):
    print(*a, file=sys.stderr, flush=True)


SEED_BASE = 0x0FD2              # payload of global frame i = splitmix64(SEED_BASE + i)


def make_workload(frame_ids, noise_seed):
    """Synthetic frames for the given global frame ids -> (iq complex64[n*PITCH], payloads uint8[n, PAYLOAD])."""
    from fun_ofdm_amd import synth
    n_frames = len(frame_ids)
    seed_base = noise_seed
    pays = synth.splitmix64_bytes(SEED_BASE, n_frames, PAYLOAD, ids=frame_ids)
    iq = np.empty(n_frames * PITCH, np.complex64)
    step = 500
    for a in range(0, n_frames, step):
        b = min(n_frames, a + step)
        fr = synth.build_frames(pays[a:b], RATE)
        part, _ = synth.make_stream(fr, PITCH, LEAD, SNR_DB, seed=seed_base * 1000003 + a)
        iq[a * PITCH:b * PITCH] = part
    return iq, pays


def cpu_baseline(iq, descs, ends, pays, budget_s=15.0):
    """The oracle (a port of the reference's per-frame path: fft_symbols..frame_decoder on one
    alignment) timed on this host's cores on a bounded sample of the same workload."""
    from oracle import pyoracle as po
    cores = os.cpu_count() or 1
    n = min(descs.size, 64 * cores)
    # size the sample from a quick probe so that the leg stays near budget_s
    t0 = time.perf_counter()
    po.decode_batch_f32(iq, descs[:cores], ends[:cores], slot_bytes=PAYLOAD, threads=cores)
    probe = time.perf_counter() - t0
    n = int(min(descs.size, max(cores, cores * budget_s / max(probe, 1e-3))))
    n_samp = int(ends[n - 1])
    t0 = time.perf_counter()
    psdu, res = po.decode_batch_f32(iq[:n_samp], descs[:n], ends[:n], slot_bytes=PAYLOAD, threads=cores)
    dt = time.perf_counter() - t0
    real = np.nonzero((descs["lts1_pos"][:n] - (LEAD + 184)) % PITCH == 0)[0]
    in_frame = real.size * (320 + 80 * 40)
    out = dict(value=in_frame / dt / 1e6, unit="Msamples/s", cores=cores, kind="port",
               sample="%d of the workload's alignments (%d frames, %d samples fed), oracle fo_decode_batch_f32 on %d threads, %.1f s"
                      % (n, real.size, n_samp, cores, dt))
    # the reference's own structure for comparison: process_samples() over six block threads + the caller,
    # 4096-sample chunks, pre-sync included -- one chain, on a few hundred frames
    try:
        nf = min(len(pays), 400)
        chain = po.ReceiverChain(threaded=True)
        t0 = time.perf_counter()
        got = chain.run_stream(iq[:nf * PITCH], chunk=4096)
        dt_c = time.perf_counter() - t0
        out["reference_structure"] = {"value": round(nf * (320 + 80 * 40) / dt_c / 1e6, 2), "unit": "Msamples/s", "threads": 7,
                                      "sample": "%d frames through the oracle's receiver_chain (frame_detector .. frame_decoder as "
                                                "six block threads, 4096-sample calls), %d payloads out, %.1f s" % (nf, len(got), dt_c)}
    except Exception as e:                                # the headline baseline above does not depend on this leg
        out["reference_structure"] = {"error": str(e)}
    return out, psdu, res, n


def pmc_traffic(kernel, frames):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/), if they were taken on
    this workload size; PMC counters cannot be read from inside the timed process."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if name.endswith("_pmc_hbm.json"):
            d = json.load(open(os.path.join(pdir, name)))
            if d.get("frames_per_gpu") == frames and kernel in d.get("kernels", {}):
                best = d["kernels"][kernel]["hbm_bytes_per_launch"]
    return best


def pmc_valu(kernel, frames):
    """VALU instructions per launch of `kernel` from the committed SQ counter pass (profiles/*_pmc_sq.json)."""
    pdir = os.path.join(ROOT, "profiles")
    best = None
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if name.endswith("_pmc_sq.json"):
            d = json.load(open(os.path.join(pdir, name)))
            if d.get("frames_per_gpu") == frames and kernel in d.get("per_launch", {}):
                best = d["per_launch"][kernel].get("SQ_INSTS_VALU")
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=10000, help="frames per GPU (BASELINE config 2: 10000)")
    ap.add_argument("--viterbi", type=int, default=2, help="0: lane per state, 1: packed + serial chain-back, 2: packed + segment chain-back")
    ap.add_argument("--tb-segment", type=int, default=0, help="viterbi 2: data steps per chain-back segment (0: library default)")
    ap.add_argument("--tb-overlap", type=int, default=-1, help="viterbi 2: run-in steps of a segment (-1: library default)")
    ap.add_argument("--frontend", type=int, default=-1, help="-1: library default, 0: wave-per-symbol, 1: lane-per-symbol, 2: quad-per-symbol kernel")
    ap.add_argument("--tx", choices=("host", "device"), default="host",
                    help="where the synthetic frames are built: numpy on the host (default) or foa_tx_* on the device")
    ap.add_argument("--no-pipeline", action="store_true", help="finish of a step on the same stream as the rest (no overlap with the next step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sync-leg", action="store_true", help="skip the extra leg with the device pre-sync (profiling: keeps its launches out of the kernel averages)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import fun_ofdm_amd as foa

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # FOA_BENCH_BACKEND=gloo lets the N>1 code path be exercised on a box with fewer GPUs than ranks (ranks then
    # share devices and the gather goes through host memory); the real runs use nccl (= RCCL over xGMI).
    backend = os.environ.get("FOA_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    cdev = dev if backend == "nccl" else torch.device("cpu")       # where collective tensors live

    t0 = time.perf_counter()
    from fun_ofdm_amd import shard, synth
    n_global = args.frames * world
    my_ids = shard.local_frame_ids(n_global, rank, world).numpy()       # global frame i -> rank i mod G
    if args.tx == "device":
        # frame_builder + channel on the device; the samples come back once for the host-side checks
        gen = foa.Receiver(dev_index)
        pays = synth.splitmix64_bytes(SEED_BASE, len(my_ids), PAYLOAD, ids=my_ids)
        d_frames = gen.tx_build_frames(torch.from_numpy(pays).to(dev), RATE)
        d_gen = gen.tx_channel(d_frames, PITCH, LEAD, SNR_DB, seed=7919 * (rank + 1))
        iq = d_gen.cpu().numpy().reshape(-1).view(np.complex64)
        del d_frames, d_gen
        gen.close()
    else:
        iq, pays = make_workload(my_ids, 7919 * (rank + 1))
    t1 = time.perf_counter()
    descs = foa.find_alignments(iq)                       # host-side frame_detector + timing_sync
    ends = foa.alignment_ends(descs, iq.size)
    t2 = time.perf_counter()
    real = np.nonzero((descs["lts1_pos"] - (LEAD + 184)) % PITCH == 0)[0]
    which = (descs["lts1_pos"][real] - (LEAD + 184)) // PITCH          # local frame index of each alignment that sits on a frame
    if rank == 0:
        log("[bench] rank0: %d frames generated in %.1f s, sync found %d alignments (%d on frames) in %.1f s"
            % (args.frames, t1 - t0, descs.size, real.size, t2 - t1))
    m = descs.size
    frame_samples = 320 + 80 * 40

    rx = foa.Receiver(dev_index)
    rx.set_option("viterbi", args.viterbi)
    if args.tb_segment > 0:
        rx.set_option("tb_segment", args.tb_segment)
    if args.tb_overlap >= 0:
        rx.set_option("tb_overlap", args.tb_overlap)
    rx.set_option("frontend", args.frontend)
    rx.set_option("pipeline", 0 if args.no_pipeline else 1)
    rx.set_option("record_soft", 0)        # PSDUs are the output; soft bytes are only kept for diagnostics
    rx.reserve(iq.size, m)
    d_iq = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).to(dev)
    d_desc = torch.from_numpy(descs.view(np.uint8).copy()).to(dev)
    d_ends = torch.from_numpy(ends).to(dev)
    # three output sets in rotation: with several ranks the PSDUs of step k-2 are gathered while step k is being queued and
    # step k-1's chain-back has yet to run, so consecutive steps must not share their output buffers
    n_out = 3 if world > 1 else 1
    out_psdu = [torch.zeros((m, PAYLOAD), dtype=torch.uint8, device=dev) for _ in range(n_out)]
    out_res = [torch.zeros((m, 4), dtype=torch.int32, device=dev) for _ in range(n_out)]
    d_psdu, d_res = out_psdu[0], out_res[0]
    d_real = torch.from_numpy(real).to(dev)
    d_which = torch.from_numpy(which).to(dev)
    gathered = [None]

    read_done = [None] * n_out

    def gather_now(i):
        # slots in local frame order (a frame the detector missed leaves a zero slot, like a CRC failure)
        buf = out_psdu[i]
        local = torch.zeros((args.frames, PAYLOAD), dtype=torch.uint8, device=dev)
        local.index_copy_(0, d_which, buf.index_select(0, d_real))
        read_done[i] = torch.cuda.Event()
        read_done[i].record()
        gathered[0] = shard.gather_psdus(local.to(cdev), n_global, rank, world)

    issued = [0]          # steps queued since the last finish_steps()
    done = [0]            # of which gathered

    def step():
        # Queue this step's front end and forward pass; the previous step's chain-back + finish runs under it on the
        # library's second stream.  With several ranks the PSDUs of the step two back are gathered meanwhile: that step is
        # complete by now, so the host is not held up and the next step's front end is queued in time.
        k = issued[0]
        if read_done[k % n_out] is not None:
            # this call will overwrite an output set that a gather has read (on torch's stream, which the library's streams
            # are not ordered against): wait for that read -- one event, queued a step ago; a synchronize of torch's stream
            # would do too, but on this runtime it waits for the library's streams as well and costs the loop 3-8 %
            read_done[k % n_out].synchronize()
            read_done[k % n_out] = None
        rx.decode_frames_dev(d_iq, d_desc, d_ends, out_psdu[k % n_out], out_res[k % n_out])
        issued[0] = k + 1
        if world > 1 and k - done[0] >= 2:
            rx.wait_age(2)
            gather_now(done[0] % n_out)
            done[0] += 1

    def finish_steps():
        rx.sync()
        while world > 1 and done[0] < issued[0]:         # the last two steps' PSDUs
            gather_now(done[0] % n_out)
            done[0] += 1
        last = (issued[0] - 1) % n_out if issued[0] else 0
        issued[0] = done[0] = 0
        return out_psdu[last], out_res[last]

    for _ in range(args.warmup):
        step()
    d_psdu, d_res = finish_steps()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    kern = {k: 0.0 for k in ("header", "scan", "symbols", "viterbi_fwd", "viterbi_finish", "total")}
    timing_age = 2 if (args.steps <= 50 and args.viterbi == 2 and not args.no_pipeline) else 1
    t_start = time.perf_counter()
    for i in range(args.steps):
        step()
        if timing_age == 2:
            if i > 1:                                    # per-kernel HIP-event times of the step two back: complete for sure,
                for k, v in rx.kernel_ms(age=2).items():  # so the host is not held up (the next call's front end must be
                    kern[k] += v                         # queued while this step's forward pass is still running)
        elif args.steps <= 50 and i > 0:
            for k, v in rx.kernel_ms(previous=True).items():
                kern[k] += v
    d_psdu, d_res = finish_steps()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if args.steps <= 50:                                 # ... and of the last step(s), after the clock has stopped
        if timing_age == 2 and args.steps > 1:
            for k, v in rx.kernel_ms(previous=True).items():
                kern[k] += v
        for k, v in rx.kernel_ms().items():
            kern[k] += v
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- extra leg (not `value`): the same pass preceded by frame_detector + timing_sync on the device ----
    with_sync = None
    if world == 1 and not args.no_sync_leg:
        cap = iq.size // 300 + 16
        s_desc = torch.zeros(cap * 48, dtype=torch.uint8, device=dev)
        s_ends = torch.zeros(cap, dtype=torch.int64, device=dev)
        s_psdu = torch.zeros((m, PAYLOAD), dtype=torch.uint8, device=dev)
        s_res = torch.zeros((m, 4), dtype=torch.int32, device=dev)
        n_found = rx.sync_dev(d_iq, s_desc, s_ends)
        torch.cuda.synchronize()
        t_s = time.perf_counter()
        for _ in range(10):
            n_found = rx.sync_dev(d_iq, s_desc, s_ends)
            rx.decode_frames_dev(d_iq, s_desc[:n_found * 48], s_ends[:n_found], s_psdu[:n_found], s_res[:n_found])
        rx.sync()
        torch.cuda.synchronize()
        dt_s = (time.perf_counter() - t_s) / 10
        same_sync = n_found == m and bool(torch.equal(s_psdu, d_psdu)) and bool(torch.equal(s_res, d_res))
        with_sync = {"Msamples_per_s": round(args.frames * frame_samples / dt_s / 1e6, 1), "ms_per_step": round(dt_s * 1e3, 4),
                     "alignments": int(n_found), "same_results_as_host_sync": same_sync}

    # ---- correctness of what was timed: every frame decodes to its payload, bit-exact ----
    res = d_res.cpu().numpy()
    psdu = d_psdu.cpu().numpy()
    ok_frames = int((res[real, 0] == 0).sum())
    okm = res[real, 0] == 0
    # every frame whose CRC passed must carry exactly the transmitted payload (a frame may legitimately fail
    # its CRC at 25 dB; the CPU receiver fails the same ones -- checked against the oracle below)
    # (the reference's detector may also miss a frame: such a frame is reported, not counted as exact or inexact)
    exact = bool(np.array_equal(psdu[real][okm], pays[which][okm])) and np.unique(which).size == real.size
    n_frames_total = n_global
    if world > 1:
        flag = torch.tensor([1 if exact else 0], dtype=torch.int32, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        exact = bool(flag.item())
        oks = torch.tensor([ok_frames], dtype=torch.int64, device=cdev)
        dist.all_reduce(oks)
        ok_frames = int(oks.item())
        if rank == 0:
            # the gathered slots are in global frame order: rows of rank 0's own frames must equal its local result
            g = gathered[0].cpu().numpy()
            exact = exact and g.shape == (n_global, PAYLOAD) and bool(np.array_equal(g[0::world][which][okm], pays[which][okm]))
            all_pays = synth.splitmix64_bytes(SEED_BASE, n_global, PAYLOAD)
            nz = g.any(axis=1)                           # frames whose CRC failed leave their slot zeroed
            exact = exact and bool(np.array_equal(g[nz], all_pays[nz])) and int(nz.sum()) == ok_frames

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        in_frame = n_frames_total * frame_samples
        value = in_frame / (elapsed / args.steps) / 1e6
        out = {
            "metric": "RX Msamples/s @20 MHz, 54 Mbps 64-QAM r=3/4; PSDU bit-exact vs CPU",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic" if args.tx == "host" else "synthetic (built on the device: foa_tx_build_frames_dev + foa_tx_channel_dev)",
            "config": {"workload": "BASELINE configs[1]: %d frames/GPU x 1024-byte PSDU payload, 64-QAM r=3/4 (54 Mbps), AWGN 25 dB"
                                   % args.frames,
                       "frames_per_gpu": args.frames, "frame_samples": frame_samples, "slot_pitch_samples": PITCH,
                       "counted_samples": "in-frame only (3520/frame)", "value_all_samples_fed": round(value * PITCH / frame_samples, 1),
                       "x_realtime_20MSps": round(value / 20.0, 1), "psdu_bit_exact": exact, "frames_ok": ok_frames, "frames_found_by_sync_rank0": int(real.size),
                       "alignments_decoded_per_gpu": m, "frontend_dtype": "f64", "viterbi_kernel": args.viterbi, "frontend_kernel": args.frontend,
                       "steps_pipelined": bool(args.viterbi == 2 and not args.no_pipeline),
                       "sharding": ("global frame i on rank i mod %d, one %s gather of PSDU slots to rank 0 per step" % (world, backend)) if world > 1 else "single GPU"},
        }
        if with_sync:
            out["config"]["incl_device_pre_sync"] = with_sync
        if args.steps <= 50:
            kms = {k: v / args.steps for k, v in kern.items()}
            # dominant kernel: the Viterbi forward pass.  Algorithmic bytes per frame: one branch-metric dword
            # in and 64 decision bits out per trellis step (39 symbols x 216 steps).
            alg_bytes = real.size * 39 * 216 * (4 + 8)
            ach = alg_bytes / (kms["viterbi_fwd"] * 1e-3) / 1e9
            fwd_kernel = {0: "k_viterbi_v1", 1: "k_viterbi_fwd2", 2: "k_viterbi_fwd3"}[args.viterbi]
            out["roofline"] = {"bound": "hbm", "kernel": fwd_kernel, "achieved": round(ach, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": round(ach / HBM_PEAK_GBPS, 5), "traffic": pmc_traffic(fwd_kernel, args.frames),
                               "algorithmic_bytes_per_launch": int(alg_bytes), "avg_kernel_ms": round(kms["viterbi_fwd"], 4),
                               "note": "bound by VALU issue, not by HBM (SURVEY 8d): see valu_issue and DESIGN.md 4"}
            if out["config"]["steps_pipelined"]:
                # consecutive forward passes run on two streams and overlap at their ends, so a launch lasts longer than a
                # step: the launch duration (what a kernel trace reports, used above) counts the shared time twice
                step_ms = elapsed / args.steps * 1e3
                out["roofline"]["launches_overlap"] = {"ms_per_step": round(step_ms, 4),
                                                       "achieved_at_step_rate": round(alg_bytes / (step_ms * 1e-3) / 1e9, 2),
                                                       "frac_at_step_rate": round(alg_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)}
            # the limit that actually binds: a wave64 VALU instruction holds its SIMD for 4 clocks -> 1024 SIMDs x 2.4 GHz / 4
            nv = pmc_valu(fwd_kernel, args.frames)
            if nv:
                peak_gi = 1024 * 2.4 / 4.0
                ach_gi = nv / (kms["viterbi_fwd"] * 1e-3) / 1e9
                out["roofline"]["valu_issue"] = {"achieved": round(ach_gi, 1), "peak": round(peak_gi, 1), "unit": "G wave-instr/s",
                                                 "frac": round(ach_gi / peak_gi, 4), "valu_instr_per_launch": int(nv),
                                                 "source": "SQ_INSTS_VALU, profiles/*_pmc_sq.json; duration live from HIP events"}
            out["kernel_ms"] = {k: round(v, 4) for k, v in kms.items()}
        if not args.no_cpu_baseline:
            cb, opsdu, ores, n_cb = cpu_baseline(iq, descs, ends, pays)
            out["cpu_baseline"] = cb
            same = bool(np.array_equal(ores.view(np.int32).reshape(-1, 4), res[:n_cb]))
            okm = res[:n_cb, 0] == 0
            same = same and bool(np.array_equal(opsdu[okm], psdu[:n_cb][okm]))
            out["config"]["gpu_equals_cpu_on_sample"] = same
            out["config"]["cpu_sample_alignments"] = int(n_cb)
            out["config"]["psdu_bit_exact"] = bool(exact and same)
        print(json.dumps(out), flush=True)
    rx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
