#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

Run in the dev container (needs /root/reference for the *_ref.npz files):

    python oracle/gen_golden.py

Provenance of each file (also written to tests/golden/MANIFEST.json):
  *_ref.npz   outputs of the REAL reference, i.e. of oracle/_ref/libfun_ofdm_ref.so, which is the
              reference's own translation units compiled where they lie (oracle/Makefile).
  frames.npz  outputs of the oracle restatement (oracle/fo_oracle.c) on seeded synthetic frames; the
              restatement itself is pinned against the *_ref fixtures and against the live _ref
              library by tests/test_oracle_vs_ref.py.
A fixture is data (inputs + expected outputs); no reference source text is stored.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
P_REF = 0.0124   # nominal frame power used for the AWGN level


def awgn(rng, n, snr_db):
    sigma = np.sqrt(P_REF / (2.0 * 10.0 ** (snr_db / 10.0)))
    return (rng.normal(size=n) + 1j * rng.normal(size=n)) * sigma


def viterbi_kat(rng):
    R = po.Ref
    syms, bits, outs = [], [], []

    def add(s, nb):
        syms.append(np.ascontiguousarray(s, np.uint8)); bits.append(nb); outs.append(R.conv_decode(s, nb))

    add(rng.integers(0, 256, 48, dtype=np.uint8), 18)                       # SIGNAL-sized garbage
    for nb in (18, 18, 42, 90, 186, 570):
        d = rng.integers(0, 256, (nb + 13) // 8 + 1, dtype=np.uint8)
        e = po.conv_encode(d, nb).astype(float) * 255
        add(np.clip(e + rng.normal(0, 50, e.size), 0, 255), nb)
    for nb in (138, 858):                                                    # pure garbage
        add(rng.integers(0, 256, 2 * (nb + 6), dtype=np.uint8), nb)
    for nb, val in ((66, 0), (66, 255), (66, 127), (210, 128)):              # constant inputs
        add(np.full(2 * (nb + 6), val, np.uint8), nb)
    for nb, punct in ((282, 3), (426, 2)):                                   # erasure patterns 3/4 and 2/3
        d = rng.integers(0, 256, (nb + 13) // 8 + 1, dtype=np.uint8)
        e = np.clip(po.conv_encode(d, nb).astype(float) * 255 + rng.normal(0, 90, 2 * (nb + 6)), 0, 255).astype(np.uint8)
        if punct == 3:
            e[2::6] = 127; e[4::6] = 127
        else:
            e[1::4] = 127
        add(e, nb)
    # a long 9 Mbps-like block (metric saturation and renormalisation are exercised constantly)
    nb = 36 * 120 - 6
    d = rng.integers(0, 256, (nb + 13) // 8 + 1, dtype=np.uint8)
    e = np.clip(po.conv_encode(d, nb).astype(float) * 255 + rng.normal(0, 70, 2 * (nb + 6)), 0, 255).astype(np.uint8)
    e[2::6] = 127; e[4::6] = 127
    add(e, nb)
    return _pack_kat(syms, bits, outs)


def viterbi_long_kat(rng):
    """Saturation / renormalisation corner cases at the trellis lengths of BASELINE configs 2 and 3,
    through the real SSE decoder: constant soft bytes, uniform garbage, alternating extremes, a codeword with every second
    pair erased, an inverted codeword.  (tests/test_gpu_parity.py feeds them to every GPU Viterbi kernel.)"""
    R = po.Ref
    syms, bits, outs = [], [], []

    def add(s, nb):
        syms.append(np.ascontiguousarray(s, np.uint8)); bits.append(nb); outs.append(R.conv_decode(s, nb))

    for nb in (8418, 32826):
        n = 2 * (nb + 6)
        for val in (0, 255, 127, 128, 1, 254):
            add(np.full(n, val, np.uint8), nb)
        add(rng.integers(0, 256, n, dtype=np.uint8), nb)
        add(np.where(np.arange(n) % 2 == 0, 0, 255), nb)
        add(np.where((np.arange(n) // 2) % 2 == 0, 0, 255), nb)
        add(np.arange(n) * 7 % 256, nb)
        d = rng.integers(0, 256, (nb + 13) // 8 + 1, dtype=np.uint8)
        code = po.conv_encode(d, nb).astype(np.uint8) * 255
        half = code.copy()
        half.reshape(-1, 2)[1::2] = 127
        add(half, nb)
        add(255 - code, nb)
        noisy = np.clip(code.astype(float) + rng.normal(0, 110, n), 0, 255).astype(np.uint8)     # marginal: many renormalisations
        noisy[2::6] = 127; noisy[4::6] = 127
        add(noisy, nb)
    return _pack_kat(syms, bits, outs)


def _pack_kat(syms, bits, outs):
    off = np.cumsum([0] + [s.size for s in syms])
    ooff = np.cumsum([0] + [o.size for o in outs])
    return dict(symbols=np.concatenate(syms), sym_off=off, data_bits=np.array(bits), decoded=np.concatenate(outs), dec_off=ooff)


def codec_kat(rng):
    R = po.Ref
    out = {}
    for r in range(po.NUM_RATES):
        rp = po.rate_params(r)
        car = (rng.normal(size=96) + 1j * rng.normal(size=96)) * 0.8
        car[:4] = [0, 1e-9, -1e-9, 3 + 3j]
        car[4:8] = np.array([1, -1, 1j, -1j]) * 1e11         # beyond int range: cvttsd2si indefinite value
        out["demod_in_%d" % r] = car
        out["demod_out_%d" % r] = R.demodulate(car, r)
        by = rng.integers(0, 256, rp["cbps"] * 2, dtype=np.uint8)
        out["bytes_in_%d" % r] = by
        out["depunct_out_%d" % r] = R.depuncture(by, r)
        bits = rng.integers(0, 2, rp["dbps"] * 2 * 2, dtype=np.uint8)
        out["bits_in_%d" % r] = bits
        out["punct_out_%d" % r] = R.puncture(bits, r)
        mb = rng.integers(0, 2, rp["cbps"], dtype=np.uint8)
        out["mod_in_%d" % r] = mb
        out["mod_out_%d" % r] = R.modulate(mb, r)
    by = rng.integers(0, 256, 96, dtype=np.uint8)
    out["il_in"] = by
    out["il_out"] = R.interleave(by)
    out["deil_out"] = R.deinterleave(by)
    d = rng.integers(0, 256, 20, dtype=np.uint8)
    out["enc_in"] = d
    out["enc_out"] = R.conv_encode(d, 130)
    car = rng.normal(size=48 * 130) + 1j * rng.normal(size=48 * 130)      # > 127 symbols: polarity wraps
    out["map_in"] = car.astype(np.complex64)
    out["map_out"] = R.symbol_map(car.astype(np.complex64).astype(np.complex128)).astype(np.complex64)
    out["preamble"] = R.table("ref_preamble_samples", 320)
    out["lts_freq"] = R.table("ref_lts_freq_domain", 64)
    out["lts_time_conj"] = R.table("ref_lts_time_domain_conj", 64)
    out["rate_table"] = np.array([[R.rate_params(r)[k] for k in ("rate_field", "cbps", "dbps", "bpsc", "rate")] for r in range(11)])
    return out


def blocks_kat(rng):
    """A short noisy two-frame stream with CFO through the real frame_detector / timing_sync /
    channel_est / phase_tracker (fft_symbols in between is the oracle's: FFTW3 is not available)."""
    R = po.Ref
    parts = []
    for i, r in enumerate((10, 2)):
        f = po.build_frame(rng.integers(0, 256, 60, dtype=np.uint8), r)
        cfo = np.exp(2j * np.pi * (2500.0 * (1 - 2 * i)) * np.arange(f.size) / 20e6) * np.exp(1j * rng.uniform(0, 6))
        parts += [np.zeros(int(rng.integers(150, 400)), complex), f * cfo]
    parts.append(np.zeros(700, complex))
    s = np.concatenate(parts)
    s = (s + awgn(rng, s.size, 25.0)).astype(np.complex64)
    n = (s.size // 1024 + 1) * 1024
    s = np.concatenate([s, np.zeros(n - s.size, np.complex64)])
    fd, ts, ce, pt = (R.Block(k) for k in ("frame_detector", "timing_sync", "channel_est", "phase_tracker"))
    fs = po.FFTSymbols()
    fd_tags, ts_tags, ts_samples, ce_in, ce_out, ce_tags, pt_out = [], [], [], [], [], [], []
    for x in range(0, n, 1024):
        a = fd.work(s[x:x + 1024].astype(np.complex128))
        b = ts.work(a)
        v = fs.work(b)
        e = ce.work(v)
        g = pt.work(e)
        fd_tags.append(a["tag"]); ts_tags.append(b["tag"]); ts_samples.append(b["sample"])
        ce_in.append(v); ce_out.append(e["samples"]); ce_tags.append(e["tag"]); pt_out.append(g["samples"])
    ce_in = np.concatenate(ce_in)
    return dict(stream=s, chunk=np.array(1024), fd_tags=np.concatenate(fd_tags).astype(np.int8),
                ts_tags=np.concatenate(ts_tags).astype(np.int8), ts_samples=np.concatenate(ts_samples),
                ce_in=ce_in["samples"], ce_in_tags=ce_in["tag"].astype(np.int8),
                ce_out=np.concatenate(ce_out), ce_out_tags=np.concatenate(ce_tags).astype(np.int8),
                pt_out=np.concatenate(pt_out))


def frames(rng):
    """Per-rate seeded frames (complex64, lead-in noise, AWGN) with the oracle's taps and results."""
    out = {}
    cases = []
    for r in range(po.NUM_RATES):
        cases.append(("rate%d" % r, r, 100 if r != 10 else 257, 25.0, 0.0))
    cases.append(("lowsnr", 10, 300, 14.0, 0.0))       # CRC failure expected
    cases.append(("len0", 0, 0, 30.0, 0.0))
    cases.append(("cfo", 8, 120, 25.0, 3000.0))
    cases.append(("sat9", 2, 700, 25.0, 0.0))          # 9 Mbps, long enough to saturate metrics
    names = []
    for name, r, ln, snr, cfo in cases:
        pay = rng.integers(0, 256, ln, dtype=np.uint8)
        f = po.build_frame(pay, r)
        f = f * np.exp(2j * np.pi * cfo * np.arange(f.size) / 20e6) * np.exp(1j * rng.uniform(0, 6))
        s = np.concatenate([np.zeros(200, complex), f, np.zeros(360, complex)])
        s = (s + awgn(rng, s.size, snr)).astype(np.complex64)
        descs = po.find_alignments_f32(s)
        assert descs.size == 1, (name, descs)
        res, psdu, taps = po.decode_alignment_f32(s, descs[0], taps=True)
        chain = po.ReceiverChain().run_stream(s.astype(np.complex128))
        assert (len(chain) == 1) == (res["status"] == po.ST_OK), name
        if psdu is not None:
            assert chain[0] == psdu.tobytes(), name
        names.append(name)
        out[name + "_iq"] = s
        out[name + "_payload"] = pay
        out[name + "_desc"] = descs
        out[name + "_res"] = np.array([res["status"], res["rate"], res["length"], res["num_symbols"]], np.int32)
        out[name + "_hinv"] = taps["hinv"].astype(np.complex64)
        out[name + "_eq"] = taps["eq"].astype(np.complex64)
        out[name + "_soft"] = taps.get("soft", np.zeros(0, np.uint8))
        out[name + "_psdu"] = psdu if psdu is not None else np.zeros(0, np.uint8)
        print("  %-8s rate %2d len %4d snr %4.1f -> status %d" % (name, r, ln, snr, res["status"]))
    # header failure: SIGNAL symbol corrupted (carriers of the SIGNAL window zeroed out)
    pay = rng.integers(0, 256, 50, dtype=np.uint8)
    f = po.build_frame(pay, 3)
    f[320:400] = 0
    s = (np.concatenate([np.zeros(200, complex), f, np.zeros(300, complex)]) + awgn(rng, f.size + 500, 25.0)).astype(np.complex64)
    descs = po.find_alignments_f32(s)
    res, psdu = po.decode_alignment_f32(s, descs[0])
    names.append("hdrfail")
    out["hdrfail_iq"] = s; out["hdrfail_desc"] = descs
    out["hdrfail_res"] = np.array([res["status"], res["rate"], res["length"], res["num_symbols"]], np.int32)
    out["hdrfail_psdu"] = np.zeros(0, np.uint8)
    print("  hdrfail -> status %d" % res["status"])
    out["names"] = np.array(names)
    return out


def frames_reftx(rng):
    """Frames whose TRANSMIT side is assembled from the REAL reference: conv_encode, puncture and interleave of oracle/_ref produce the coded
    bits (ppdu.cpp:115-165's order), and the real modulate / symbol_map are asserted equal to the oracle's on these very bits.  All eleven
    rates at 1 and 4095 payload bytes.  Stored: payload, rate, the real pieces' interleaved coded bits (packed) -- a few KB per frame; the
    tests rebuild the samples from the bits with the oracle's modulate / symbol_map / inverse DFT / preamble (pinned elsewhere against
    the real ones; fft.cpp itself cannot be built here) and expect the PAYLOAD back, from the oracle's receiver on the CPU and from the
    device.  What only reading pins after this: the header-field layout, the scrambler and where the CRC goes (ppdu.cpp:75-147)."""
    R = po.Ref
    out = {}
    names = []
    for r in range(po.NUM_RATES):
        rp = po.rate_params(r)
        for ln in (1, 4095):
            pay = rng.integers(0, 256, ln, dtype=np.uint8)
            nsym = po.num_symbols(r, ln)
            nbits = nsym * rp["dbps"]
            data = np.zeros(nbits // 8 + 1, np.uint8)
            data[2:2 + ln] = pay
            crc = po.crc32(data[:2 + ln])
            data[2 + ln:6 + ln] = np.frombuffer(np.uint32(crc).tobytes(), np.uint8)
            scr = np.zeros_like(data)
            scr[:nbits // 8] = po.scramble(data[:nbits // 8])
            enc = R.conv_encode(scr, nbits - 6)                      # REAL viterbi::conv_encode
            inter = R.interleave(R.puncture(enc, r))                 # REAL puncturer::puncture, interleaver::interleave
            car = R.modulate(inter, r)                               # REAL modulator::modulate
            assert np.array_equal(car, po.modulate(inter, r)) and np.array_equal(car, po.encode_data(pay, r))
            bins = R.symbol_map(np.concatenate([po.encode_header(r, ln), car]))      # REAL symbol_mapper::map
            assert np.array_equal(bins, po.symbol_map(np.concatenate([po.encode_header(r, ln), car])))
            frame = po.frame_from_coded_bits(inter, r, ln)
            assert np.array_equal(frame, po.build_frame(pay, r))
            name = "r%d_len%d" % (r, ln)
            names.append(name)
            out[name + "_payload"] = pay
            out[name + "_bits"] = np.packbits(inter.astype(np.uint8))
            out[name + "_nbits"] = np.array([inter.size], np.int64)
            print("  %-12s %6d coded bits, %6d samples" % (name, inter.size, frame.size))
    out["names"] = np.array(names)
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    po.build(ref=True)
    only = set(sys.argv[1:])                 # optional: regenerate just the named files
    mpath = os.path.join(OUT, "MANIFEST.json")
    manifest = json.load(open(mpath)) if (only and os.path.exists(mpath)) else {}
    for fname, fn, seed, prov in (
        ("viterbi_ref.npz", viterbi_kat, 101, "real reference viterbi::conv_decode (SSE) via oracle/_ref"),
        ("viterbi_long_ref.npz", viterbi_long_kat, 105, "real reference viterbi::conv_decode (SSE) via oracle/_ref: corner cases at 8418 / 32826 data bits"),
        ("codec_ref.npz", codec_kat, 102, "real reference modulator/interleaver/puncturer/viterbi::conv_encode/symbol_mapper/tables via oracle/_ref"),
        ("blocks_ref.npz", blocks_kat, 103, "real reference frame_detector/timing_sync/channel_est/phase_tracker via oracle/_ref (fft_symbols: oracle)"),
        ("frames.npz", frames, 104, "oracle restatement (fo_oracle.c), itself pinned against the *_ref fixtures"),
        ("frames_reftx.npz", frames_reftx, 106, "transmit side from the real reference's conv_encode / puncture / interleave (coded bits stored), modulate and symbol_map "
                                               "asserted equal to the oracle's on them; all 11 rates at 1 and 4095 bytes; expected output = the payload"),
    ):
        if only and fname not in only:
            continue
        print(fname)
        data = fn(np.random.default_rng(seed))
        np.savez_compressed(os.path.join(OUT, fname), **data)
        manifest[fname] = dict(seed=seed, provenance=prov, generator="oracle/gen_golden.py")
    with open(mpath, "w") as f:
        json.dump(manifest, f, indent=1)
    print({f: os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT)})


if __name__ == "__main__":
    main()
