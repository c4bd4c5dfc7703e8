#!/usr/bin/env python3
"""What the FIRST call of each kind costs a fresh process (the runtime loads code objects and sizes its pools under it), against the ones
after: pre-sync, decode call, and -- through a Stream -- the first batches of process_samples' engine.  One JSON line."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import fun_ofdm_amd as foa
    from fun_ofdm_amd import synth
    t_start = time.perf_counter()
    rx = foa.Receiver(0)
    t_create = time.perf_counter() - t_start
    dev = torch.device("cuda", 0)
    pays = synth.splitmix64_bytes(7, 32, 1024)
    t0 = time.perf_counter(); frames = rx.tx_build_frames(torch.from_numpy(pays).to(dev), 10); torch.cuda.synchronize(); t_tx = time.perf_counter() - t0
    d_iq = rx.tx_channel(frames, frames.shape[1] + 160, 80, 25.0, seed=3)
    torch.cuda.synchronize()
    cap = d_iq.shape[0] // 512 + 64
    d_desc = torch.zeros(cap * 48, dtype=torch.uint8, device=dev); d_end = torch.zeros(cap, dtype=torch.int64, device=dev)
    sync_ms = []
    for _ in range(3):
        t0 = time.perf_counter(); m = rx.sync_dev(d_iq, d_desc, d_end); sync_ms.append(round((time.perf_counter() - t0) * 1e3, 3))
    d_psdu = torch.zeros((m, 1024), dtype=torch.uint8, device=dev); d_res = torch.zeros((m, 4), dtype=torch.int32, device=dev)
    dec_ms = []
    for _ in range(4):
        t0 = time.perf_counter(); rx.decode_frames_dev(d_iq, d_desc[:m * 48], d_end[:m], d_psdu, d_res); rx.sync(); dec_ms.append(round((time.perf_counter() - t0) * 1e3, 3))
    iq = d_iq.cpu().numpy().reshape(-1).view(np.complex64)
    t0 = time.perf_counter(); st = foa.Stream(rx, 4096, 2); t_open = time.perf_counter() - t0
    push_ms, got = [], 0
    t_first = None
    ts = time.perf_counter()
    for o in range(0, min(iq.size, 40 * 4096), 4096):
        t0 = time.perf_counter(); got += len(st.push(iq[o: o + 4096])); push_ms.append(round((time.perf_counter() - t0) * 1e3, 3))
        if got and t_first is None:
            t_first = time.perf_counter() - ts
        time.sleep(0.0002)
    t0 = time.perf_counter(); got += len(st.flush()); t_flush = time.perf_counter() - t0
    st.close()
    print(json.dumps({"create_ms": round(t_create * 1e3, 1), "first_tx_ms": round(t_tx * 1e3, 1), "sync_ms": sync_ms, "decode_ms": dec_ms, "alignments": int(m),
                      "stream_open_ms": round(t_open * 1e3, 1), "first_payload_after_ms": None if t_first is None else round(t_first * 1e3, 2), "payloads": got, "stream_push_ms_first10": push_ms[:10], "stream_push_ms_max": max(push_ms), "stream_flush_ms": round(t_flush * 1e3, 2)}))
    rx.close()


if __name__ == "__main__":
    main()
