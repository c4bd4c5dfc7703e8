"""Pin the oracle restatement against the LIVE partial build of the real reference
(oracle/_ref, compiled from /root/reference where it lies).  Dev container only: skipped where the
reference tree is absent (the GPU box)."""
import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.skipif(not po.ref_available(), reason="reference tree not present")
R = po.Ref


def test_struct_layouts():
    assert po.ref().ref_sizeof_tagged_sample() == po.tagged_sample.itemsize == 24
    assert po.ref().ref_sizeof_tagged_vector64() == po.tagged_vec64.itemsize == 1032
    assert po.ref().ref_sizeof_tagged_vector48() == po.tagged_vec48.itemsize == 776


def test_viterbi_random_blocks():
    rng = np.random.default_rng(11)
    for it in range(400):
        nb = int(rng.integers(1, 700)) * 2
        mode = it % 4
        if mode == 0:
            s = rng.integers(0, 256, 2 * (nb + 6), dtype=np.uint8)
        elif mode == 3:
            s = np.full(2 * (nb + 6), rng.integers(0, 256), np.uint8)
        else:
            d = rng.integers(0, 256, (nb + 13) // 8 + 1, dtype=np.uint8)
            e = po.conv_encode(d, nb).astype(float) * 255
            s = np.clip(e + rng.normal(0, 60 * mode, e.size), 0, 255).astype(np.uint8)
            if mode == 2:
                s[rng.random(s.size) < 0.33] = 127
        assert np.array_equal(po.conv_decode(s, nb), R.conv_decode(s, nb)), (it, nb)


def test_simd_forward_pass_against_the_compiled_reference_decoder():
    """The oracle's SSE forward pass (the TIMED CPU baseline) + chain-back against src/viterbi.cpp compiled in place."""
    rng = np.random.default_rng(31)
    for nb in (18, 100, 8418, 20002):
        n = nb + 6
        for s in (rng.integers(0, 256, 2 * n, dtype=np.uint8), rng.choice(np.array([0, 255, 127], np.uint8), 2 * n),
                  (po.conv_encode(rng.integers(0, 256, nb // 8 + 2, dtype=np.uint8), nb).astype(np.uint8) * 255)):
            dec, _ = po.viterbi_forward_simd(s, n)
            assert np.array_equal(po.viterbi_chainback(dec, nb), po.Ref.conv_decode(s, nb)), nb


def test_viterbi_long_punctured_blocks():
    # 4092-byte 9 Mbps shape: 32 796 trellis steps, erasures every third pair
    rng = np.random.default_rng(12)
    for it in range(3):
        nb = 36 * 911 - 6
        d = rng.integers(0, 256, (nb + 13) // 8 + 1, dtype=np.uint8)
        e = np.clip(po.conv_encode(d, nb).astype(float) * 255 + rng.normal(0, 75, 2 * (nb + 6)), 0, 255).astype(np.uint8)
        e[2::6] = 127
        e[4::6] = 127
        assert np.array_equal(po.conv_decode(e, nb), R.conv_decode(e, nb))


def test_codec_random():
    rng = np.random.default_rng(13)
    for r in range(po.NUM_RATES):
        rp = po.rate_params(r)
        for scale in (0.05, 1.0, 4.0, 1e12):
            car = (rng.normal(size=48 * 3) + 1j * rng.normal(size=48 * 3)) * scale
            assert np.array_equal(po.demodulate(car, r), R.demodulate(car, r)), (r, scale)
        by = rng.integers(0, 256, rp["cbps"] * 5, dtype=np.uint8)
        assert np.array_equal(po.deinterleave(by), R.deinterleave(by))
        assert np.array_equal(po.interleave(by), R.interleave(by))
        assert np.array_equal(po.depuncture(by, r), R.depuncture(by, r))
        bits = rng.integers(0, 2, rp["dbps"] * 2 * 5, dtype=np.uint8)
        assert np.array_equal(po.puncture(bits, r), R.puncture(bits, r))
        mb = rng.integers(0, 2, rp["cbps"] * 3, dtype=np.uint8)
        assert np.array_equal(po.modulate(mb, r), R.modulate(mb, r))
    for x in range(0, 1 << 20, 4099):
        assert po.lib().fo_parity(x) == po.ref().ref_parity(x)


def test_tx_through_real_encoder_pieces_decodes_in_oracle():
    """TX assembled from the REAL reference's conv_encode/puncture/interleave/modulate/symbol_map
    (ppdu.cpp:115-165 order) is decoded by the oracle's receive chain: pins the orchestration of the
    files that cannot be built here (ppdu.cpp, frame_builder.cpp, frame_decoder.cpp)."""
    rng = np.random.default_rng(14)
    for r in range(po.NUM_RATES):
        rp = po.rate_params(r)
        pay = rng.integers(0, 256, 90, dtype=np.uint8)
        nsym = po.num_symbols(r, pay.size)
        nbits = nsym * rp["dbps"]
        data = np.zeros(nbits // 8 + 1, np.uint8)
        data[2:2 + pay.size] = pay
        crc = po.crc32(data[:2 + pay.size])
        data[2 + pay.size:6 + pay.size] = np.frombuffer(np.uint32(crc).tobytes(), np.uint8)
        scr = np.zeros_like(data)
        scr[:nbits // 8] = po.scramble(data[:nbits // 8])
        enc = R.conv_encode(scr, nbits - 6)
        car = R.modulate(R.interleave(R.puncture(enc, r)), r)
        bins = R.symbol_map(np.concatenate([po.encode_header(r, pay.size), car]))
        assert np.array_equal(car, po.encode_data(pay, r))
        td = np.concatenate([np.concatenate([po.ifft64(b)[48:], po.ifft64(b)]) for b in bins.reshape(-1, 64)])
        frame = np.concatenate([R.table("ref_preamble_samples", 320), td])
        assert np.array_equal(frame, po.build_frame(pay, r))
        s = np.concatenate([np.zeros(256, complex), frame, np.zeros(600, complex)])
        assert po.ReceiverChain().run_stream(s) == [pay.tobytes()]


def test_blocks_on_noisy_cfo_stream():
    rng = np.random.default_rng(15)
    parts = []
    for i, r in enumerate((10, 0, 5, 8, 9)):
        f = po.build_frame(rng.integers(0, 256, 150, dtype=np.uint8), r)
        f = f * np.exp(2j * np.pi * 1500.0 * (i - 2) * np.arange(f.size) / 20e6) * np.exp(1j * rng.uniform(0, 6))
        parts += [np.zeros(int(rng.integers(0, 700)), complex), f]
    parts.append(np.zeros(3000, complex))
    s = np.concatenate(parts)
    s = s + (rng.normal(size=s.size) + 1j * rng.normal(size=s.size)) * np.sqrt(0.0124 / 2 / 10 ** 2.2)
    s = s.astype(np.complex64).astype(np.complex128)
    n = (s.size // 4096 + 1) * 4096
    s = np.concatenate([s, np.zeros(n - s.size)])
    fd_o, ts_o, fs, ce_o, pt_o, dec = po.FrameDetector(), po.TimingSync(), po.FFTSymbols(), po.ChannelEst(), po.PhaseTracker(), po.FrameDecoder()
    fd_r, ts_r, ce_r, pt_r = (R.Block(k) for k in ("frame_detector", "timing_sync", "channel_est", "phase_tracker"))
    got = []
    for x in range(0, n, 4096):
        a, b = fd_o.work(s[x:x + 4096]), fd_r.work(s[x:x + 4096])
        assert np.array_equal(a["tag"], b["tag"]) and np.array_equal(a["sample"], b["sample"])
        c, d = ts_o.work(a), ts_r.work(b)
        assert np.array_equal(c["tag"], d["tag"]) and np.array_equal(c["sample"], d["sample"])
        v = fs.work(c)
        e, f = ce_o.work(v), ce_r.work(v)
        assert np.array_equal(e["tag"], f["tag"]) and np.array_equal(e["samples"], f["samples"], equal_nan=True)
        g, h = pt_o.work(e), pt_r.work(f)
        assert np.array_equal(g["tag"], h["tag"]) and np.array_equal(g["samples"], h["samples"], equal_nan=True)
        got += dec.work(g)
    assert len(got) == 5


def test_timing_sync_drops_the_frame_at_a_call_boundary_like_the_real_one():
    """timing_sync.cpp:99 (`if(lts_offset < 0) break;`) in the COMPILED reference: a frame whose STS_END tag is the first sample a
    4096-sample call walks over, and a sample late, is dropped -- and found again when the same stream is shifted by four samples.
    The oracle's blocks agree tag for tag; this is the behaviour the library's pre-sync reproduces by stream index
    (tests/test_synth.py, tests/test_gpu_parity.py)."""
    import fun_ofdm_amd as foa
    from fun_ofdm_amd import synth
    pays = synth.splitmix64_bytes(31, 3, 100)
    iq, _ = synth.make_stream(synth.build_frames(pays, 5), pitch=3000, lead=500, snr_db=25.0, seed=8)
    base = foa.find_alignments(iq, call=0)
    x = int(base["rot_start"][1])
    rel = int(base["lts1_pos"][1]) - 24 + 32 - x
    assert base.size == 3 and 0 < rel < 32
    found = {}
    for late in (0, 4 + 32):
        pad = (-(x + 160) + late) % 4096
        s = np.concatenate([np.zeros(pad), iq.astype(np.complex128)])
        n = (s.size // 4096 + 2) * 4096
        s = np.concatenate([s, np.zeros(n - s.size)])
        fd_o, ts_o, fd_r, ts_r = po.FrameDetector(), po.TimingSync(), R.Block("frame_detector"), R.Block("timing_sync")
        lts1 = 0
        for c in range(0, n, 4096):
            a, b = fd_o.work(s[c:c + 4096]), fd_r.work(s[c:c + 4096])
            assert np.array_equal(a["tag"], b["tag"])
            u, v = ts_o.work(a), ts_r.work(b)
            assert np.array_equal(u["tag"], v["tag"]) and np.array_equal(u["sample"], v["sample"])
            lts1 += int(np.count_nonzero(v["tag"] == 4))               # LTS1 tags written by the real timing_sync
        found[late] = lts1
        assert po.find_alignments_f32(s.astype(np.complex64)).size == lts1
    assert found == {0: 2, 36: 3}
