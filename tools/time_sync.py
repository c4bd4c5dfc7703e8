"""Times foa_rx_sync_dev alone (blocking call, stream resident) on a synthetic config-2-like stream and checks its descriptors
against the cross-check build's direct-sum flag kernel (option sync_flags 0).  Usage: python3 tools/time_sync.py [frames [product|host]]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fun_ofdm_amd as foa
from fun_ofdm_amd import synth

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dev = torch.device("cuda", 0)
only_product = len(sys.argv) > 2 and sys.argv[2] == "product"      # (under rocprofv3: one library only)
rx = foa.Receiver(0)
rxx = None if only_product else foa.Receiver(0, xcheck=True)
pays = synth.splitmix64_bytes(1, frames, 1024)
fr = rx.tx_build_frames(torch.from_numpy(pays).to(dev), 10)
s = fr.shape[1]
iq = rx.tx_channel(fr, 4096, 176, 25.0, seed=3)
n = iq.numel() // 2
cap = n // 300 + 16
out = {}
for name, r, opt in (("grouped", rx, None), ("direct(xcheck)", rxx, 0), ("grouped(xcheck lib)", rxx, 1)):
    if r is None:
        continue
    if opt is not None:
        r.set_option("sync_flags", opt)
    d = torch.zeros(cap * 48, dtype=torch.uint8, device=dev); e = torch.zeros(cap, dtype=torch.int64, device=dev)
    for _ in range(3):
        m = r.sync_dev(iq, d, e)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        m = r.sync_dev(iq, d, e)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    out[name] = (m, d[:m * 48].clone(), e[:m].clone())
    print("%-20s %d samples: %.3f ms per call, %d alignments" % (name, n, dt * 1e3, m))
if only_product:
    sys.exit(0)
a, b = out["grouped"], out["direct(xcheck)"]
print("same descriptors and ends as the direct-sum kernel:", a[0] == b[0] and bool(torch.equal(a[1], b[1])) and bool(torch.equal(a[2], b[2])))
if len(sys.argv) > 2 and sys.argv[2] == "host":                        # the host restatement (= the reference's decisions) over the same stream
    h_iq = iq.cpu().numpy().reshape(-1).view(np.complex64)
    t0 = time.perf_counter(); want = foa.find_alignments(h_iq); th = time.perf_counter() - t0
    got = a[1].cpu().numpy().view(foa.frame_desc_dtype)
    print("host pre-sync %.2f s; same alignments: %s; largest phasor difference %.2e" % (th, got.size == want.size and np.array_equal(got["lts1_pos"], want["lts1_pos"])
          and np.array_equal(got["rot_start"], want["rot_start"]), max(np.abs(got[k] - want[k]).max() for k in ("c", "s", "c_prev", "s_prev"))))
