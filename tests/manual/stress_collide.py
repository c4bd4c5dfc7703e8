"""Colliding frames and false alarms inside frames: the ORDERED PAYLOAD LIST of every path against the oracle's BLOCK-LEVEL receiver chain.

The reference's fft_symbols pushes a partly filled vector when an LTS1 tag arrives mid-symbol (fft_symbols.cpp:41-50), channel_est
equalises it with the estimate still in force (channel_est.cpp:77-81), frame_decoder copies it -- and the SIGNAL symbol of the new
alignment after it -- into the frame it is collecting (frame_decoder.cpp:52-68).  The oracle's block chain (oracle/fo_oracle.c,
fo_fft_symbols_work .. fo_frame_decoder_work; po.ReceiverChain) models exactly that; this script builds streams that exercise it:

  * a second frame laid over a first one at an arbitrary sample offset -- uniform over the first frame, or concentrated on its last two
    symbols, where the partial vector can COMPLETE the first frame -- weaker / equal / stronger, with a valid or a garbled SIGNAL;
  * bare preambles (STS + LTS, nothing behind them) inside a frame: timing_sync's false alarm;
  * a frame cut short by the next one; three frames in a pile-up.

Modes:
  cpu   (no GPU): the oracle's batch restatement (po.decode_batch_f32 over po.find_alignments_f32) against the block chain
  gpu   the batch path (host pre-sync + foa_rx_decode_frames_host; device pre-sync + foa_rx_decode_frames_dev), the stream engine
        (foa.Stream) and, with --cpp, the three process_samples modes of fun_amd::receiver_chain (examples/foa_sim), all against the block chain
Usage: python3 tests/manual/stress_collide.py cpu|gpu [first seed] [last seed] [--cpp]"""
import os
import subprocess
import sys
import tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import pyoracle as po

RATES = (0, 2, 3, 5, 6, 8, 9, 10)


def frame(rng, seed, i, length=None, rate=None):
    rate = int(rng.choice(RATES)) if rate is None else rate
    length = int(rng.integers(1, 500)) if length is None else length
    pay = np.random.default_rng(seed * 1000 + i).integers(0, 256, length, dtype=np.uint8)
    return po.build_frame(pay, rate), rate, length


def make_stream(seed):
    """One stream: a few scenes one after another, each a first frame with something laid over it."""
    rng = np.random.default_rng(seed)
    scenes = []
    for i in range(int(rng.integers(1, 5))):
        kind = rng.choice(("late", "last2", "uniform", "bare", "cut", "pile", "none"), p=(0.3, 0.2, 0.12, 0.16, 0.08, 0.08, 0.06))
        if kind == "late":
            # the case in which the partial vector could COMPLETE the first frame: a robust first frame whose last symbol is mostly padding,
            # a much stronger second preamble (the detector needs |B|^2 > 9 |A|^2 to see a plateau over A) whose LTS1 tag falls late in it
            a, rate, length = frame(rng, seed, 10 * i, length=int(rng.integers(1, 150)), rate=int(rng.choice((0, 2, 3))))
        else:
            a, rate, length = frame(rng, seed, 10 * i)
        seg = np.zeros(a.size + 6000, complex)
        seg[:a.size] += a * np.exp(1j * rng.uniform(0, 6.28))
        end_a = a.size

        def lay(off, sig, gain):
            sig = sig * gain * np.exp(1j * rng.uniform(0, 6.28))
            if rng.random() < 0.3:
                sig = sig * np.exp(2j * np.pi * rng.uniform(-4000, 4000) * np.arange(sig.size) / 20e6)
            n = min(sig.size, seg.size - off)
            seg[off:off + n] += sig[:n]
            return off + n

        gain = float(rng.choice((0.3, 1.0, 3.0, 10.0))) * 10 ** rng.uniform(-0.1, 0.1)
        if kind == "late":
            b = po.preamble_samples() if rng.random() < 0.5 else frame(rng, seed, 10 * i + 1)[0]
            lts1 = end_a - 72 + int(rng.integers(40, 72))            # the last symbol's window starts 72 before the frame's end (windows are taken 8 early)
            stop = lay(max(lts1 - 184, 0), b, 10 ** rng.uniform(0.5, 1.5))          # timing_sync tags LTS1 184 samples into a preamble
        elif kind in ("last2", "uniform", "pile"):
            b, _, _ = frame(rng, seed, 10 * i + 1)
            if rng.random() < 0.35:                                  # garbled SIGNAL
                b = b.copy()
                b[320:400] = b[320:400][::-1] * 1j
            # where B's LTS1 (its sample 192) falls relative to A: over A's last two symbols, or anywhere in A
            if kind == "last2":
                lts1 = end_a - int(rng.integers(1, 170))
            else:
                lts1 = int(rng.integers(400, end_a + 40))
            off = max(lts1 - 192, 0)
            stop = lay(off, b, gain)
            if kind == "pile":
                c, _, _ = frame(rng, seed, 10 * i + 2)
                stop = max(stop, lay(off + int(rng.integers(100, 1200)), c, gain * float(rng.choice((0.5, 1.0, 3.0)))))
        elif kind == "bare":
            pre = po.preamble_samples()
            for _ in range(int(rng.integers(1, 3))):
                lts1 = end_a - int(rng.integers(1, 170)) if rng.random() < 0.6 else int(rng.integers(400, end_a))
                lay(max(lts1 - 192, 0), pre, gain)
            stop = end_a
        elif kind == "cut":
            cut = int(rng.integers(400, end_a - 1))
            seg[cut:end_a] = 0
            b, _, _ = frame(rng, seed, 10 * i + 1)
            stop = lay(cut + (0 if rng.random() < 0.5 else int(rng.integers(1, 100))), b, 1.0)
        else:
            stop = end_a
        stop = max(stop, end_a) + (0 if rng.random() < 0.3 else int(rng.integers(1, 900)))
        scenes.append(seg[:stop])
    s = np.concatenate([np.zeros(int(rng.integers(200, 700)), complex)] + scenes + [np.zeros(int(rng.integers(300, 900)), complex)])
    snr = rng.uniform(12.0, 32.0)
    s = s + (rng.normal(size=s.size) + 1j * rng.normal(size=s.size)) * np.sqrt(0.0124 / 2 / 10 ** (snr / 10))
    return s.astype(np.complex64)


def batch_list(psdu, res):
    return [psdu[f, :res[f]["length"]].tobytes() for f in range(res.size) if res[f]["status"] == 0]


def ends_of(descs, n):
    return np.append(descs["lts1_pos"][1:], n).astype(np.int64)


def run_cpu(lo, hi, verbose=True):
    bad = tot = n_al = 0
    for seed in range(lo, hi):
        s = make_stream(seed)
        want = po.ReceiverChain().run_stream(s.astype(np.complex128))
        descs = po.find_alignments_f32(s)
        psdu, res = po.decode_batch_f32(s, descs, ends_of(descs, s.size))
        got = batch_list(psdu, res)
        tot += len(want)
        n_al += descs.size
        if got != want:
            bad += 1
            if verbose:
                print("DIFF seed", seed, "samples", s.size, "alignments", descs.size, "chain", len(want), "payloads, batch restatement", len(got),
                      "statuses", res["status"].tolist())
    return tot, n_al, bad


def run_gpu(lo, hi, cpp=False):
    import fun_ofdm_amd as foa
    import torch
    rx = foa.Receiver(0)
    exe = None
    tmp = tempfile.mkdtemp(prefix="foa_collide_")
    if cpp:
        exe = os.path.join(tmp, "foa_sim")
        libdir = os.path.dirname(foa.library_path())
        subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "examples", "foa_sim.cpp"), "-I", os.path.join(ROOT, "include"), "-L", libdir,
                        "-lfun_ofdm_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", exe], check=True)
    bad = {}
    tot = n_al = 0

    def note(path, seed, want, got):
        bad[path] = bad.get(path, 0) + 1
        print("DIFF", path, "seed", seed, "chain", len(want), "payloads, got", len(got))

    for seed in range(lo, hi):
        s = make_stream(seed)
        rng = np.random.default_rng(seed + 77)
        want = po.ReceiverChain().run_stream(s.astype(np.complex128))
        tot += len(want)
        # batch path, host pre-sync
        descs = foa.find_alignments(s)
        n_al += descs.size
        if descs.tobytes() != po.find_alignments_f32(s).tobytes():
            note("host pre-sync descriptors", seed, want, [])
        ends = foa.alignment_ends(descs, s.size)
        psdu, res = rx.decode_frames_host(s, descs, ends)
        opsdu, ores = po.decode_batch_f32(s, descs, ends)
        if not (np.array_equal(res.view(np.int32), ores.view(np.int32)) and batch_list(psdu, res) == batch_list(opsdu, ores)):
            note("batch path vs the oracle's batch restatement", seed, batch_list(opsdu, ores), batch_list(psdu, res))
        if batch_list(psdu, res) != want:
            note("batch path (host pre-sync)", seed, want, batch_list(psdu, res))
        # batch path, device pre-sync
        d_iq = torch.from_numpy(s.view(np.float32).reshape(-1, 2).copy()).to("cuda:0")
        cap = s.size // 300 + 64
        d_desc = torch.zeros(cap * 48, dtype=torch.uint8, device="cuda:0")
        d_end = torch.zeros(cap, dtype=torch.int64, device="cuda:0")
        m = rx.sync_dev(d_iq, d_desc, d_end)
        dd = d_desc.cpu().numpy()[:m * 48].view(foa.frame_desc_dtype)
        # positions exactly, phasors to 1e-12: atan2 / cos / sin come from the device's math library there (a floating-point intermediate;
        # the host restatement equals the oracle byte for byte, above)
        if not (m == descs.size and np.array_equal(dd["lts1_pos"], descs["lts1_pos"]) and np.array_equal(dd["rot_start"], descs["rot_start"]) and
                (m == 0 or max(np.abs(dd[k] - descs[k]).max() for k in ("c", "s", "c_prev", "s_prev")) < 1e-12)):
            note("device pre-sync descriptors", seed, want, [])
        if m:
            d_psdu = torch.zeros((m, 4096), dtype=torch.uint8, device="cuda:0")
            d_res = torch.zeros((m, 4), dtype=torch.int32, device="cuda:0")
            rx.decode_frames_dev(d_iq, d_desc[:m * 48], d_end[:m], d_psdu, d_res)
            rx.sync()
            r, p = d_res.cpu().numpy(), d_psdu.cpu().numpy()
            got = [p[f, :r[f, 2]].tobytes() for f in range(m) if r[f, 0] == 0]
        else:
            got = []
        if got != want:
            note("batch path (device pre-sync)", seed, want, got)
        # stream engine
        st = foa.Stream(rx, int(rng.choice((4096, 8192, 65536))), int(rng.integers(0, 3)))
        got, i = [], 0
        while i < s.size:
            n = int(rng.integers(1, 20000))
            got += st.push(s[i:i + n])
            i += n
        got += st.flush()
        st.close()
        if got != want:
            note("stream engine", seed, want, got)
        if exe:
            src, out = os.path.join(tmp, "cap.fc32"), os.path.join(tmp, "psdus")
            s.tofile(src)
            for name, extra in (("process_samples synchronous", []), ("process_samples async", ["--async", "3"]),
                                ("process_samples device", ["--device-batch", "8192", "--narrow-threads", "1"])):
                r = subprocess.run([exe, src, "--format", "fc32", "--out", out, "--chunk", str(int(rng.choice((1000, 4096, 16384))))] + extra,
                                   capture_output=True, text=True, timeout=300)
                raw, recs, o = open(out, "rb").read() if r.returncode == 0 else b"", [], 0
                while o < len(raw):
                    n = int.from_bytes(raw[o:o + 4], "little")
                    recs.append(raw[o + 4:o + 4 + n])
                    o += 4 + n
                if r.returncode != 0 or recs != want:
                    note(name, seed, want, recs)
    rx.close()
    return tot, n_al, bad


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "cpu"
    nums = [a for a in sys.argv[2:] if not a.startswith("--")]
    lo = int(nums[0]) if nums else 0
    hi = int(nums[1]) if len(nums) > 1 else 200
    if mode == "cpu":
        tot, n_al, bad = run_cpu(lo, hi)
        print("seeds %d..%d: %d alignments, %d payloads from the block chain; streams whose payload list differs (batch restatement): %d" % (lo, hi - 1, n_al, tot, bad))
    else:
        tot, n_al, bad = run_gpu(lo, hi, "--cpp" in sys.argv)
        print("seeds %d..%d: %d alignments, %d payloads from the block chain; streams that differ, by path: %s" % (lo, hi - 1, n_al, tot, bad or "none"))
