"""Parity of the HIP path (through the C ABI) against the oracle and the golden vectors.  GPU only."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4     # north_star: FFT / equaliser intermediates within 1e-4 relative (max-norm per symbol)


@pytest.fixture(scope="module")
def rx():
    import fun_ofdm_amd as foa
    r = foa.Receiver(0)
    yield r
    r.close()


def test_library_holds_one_implementation_of_every_stage():
    """The library holds ONE Viterbi path and ONE front end (the cross-check kernels of rounds 1-4 are gone: the oracle and the compiled
    reference decoder pin the product directly); the options that used to select them are unknown, not silently accepted."""
    import fun_ofdm_amd as foa
    r = foa.Receiver(0)
    try:
        for name, value in (("viterbi", 2), ("viterbi", 0), ("frontend", 2), ("frontend", -1), ("sync_flags", 1), ("forward", 3), ("lanes", 1)):
            with pytest.raises(foa.FoaError):
                r.set_option(name, value)
    finally:
        r.close()


# Chain-back segmentations (tb_segment, tb_overlap) of the Viterbi path: default, no run-in at all (most segments get re-walked),
# short segments, one long segment.  (The leading 2 is the kernel generation; rounds 1-4 kept two older ones beside it.)
VITERBI_KINDS = [(2, 960, 96), (2, 96, 0), (2, 192, 96), (2, 3072, 0)]


def _kind_id(k):
    return "v%d-S%d-L%d" % k


def _set_viterbi(rx, kind):
    rx.set_option("tb_segment", kind[1])
    rx.set_option("tb_overlap", kind[2])


def _ends(descs, n):
    e = np.empty(descs.size, np.int64)
    e[:-1] = descs["lts1_pos"][1:]
    e[-1] = n
    return e


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_library_is_native_and_loaded():
    import fun_ofdm_amd as foa
    assert foa.lib().foa_version() == 110
    assert foa.lib().foa_device_count() >= 1


@pytest.mark.parametrize("kind", VITERBI_KINDS, ids=_kind_id)
def test_conv_decode_matches_reference_sse_vectors(rx, golden, kind):
    """The Viterbi kernels of the batch path (foa_conv_decode runs k_viterbi_fwd3 / k_tb_walk / k_tb_finish on records built on
    the host), for several chain-back segmentations, fed the bytes the REAL reference decoder (viterbi.cpp:208-457,
    :108-146, compiled SSE) was fed: bit-exact, garbage / constant / erasure inputs included."""
    _set_viterbi(rx, kind)
    g = golden.viterbi_ref
    for i, nb in enumerate(g["data_bits"]):
        s = g["symbols"][g["sym_off"][i]:g["sym_off"][i + 1]]
        want = g["decoded"][g["dec_off"][i]:g["dec_off"][i + 1]]
        got = rx.conv_decode(s, int(nb))[0]
        assert np.array_equal(got, want), "KAT %d (data_bits %d)" % (i, nb)
    gl = golden.viterbi_long_ref            # corner cases at the trellis lengths of configs 2 and 3, real SSE outputs
    for i, nb in enumerate(gl["data_bits"]):
        s = gl["symbols"][gl["sym_off"][i]:gl["sym_off"][i + 1]]
        got = rx.conv_decode(s, int(nb))[0]
        assert np.array_equal(got, gl["decoded"][gl["dec_off"][i]:gl["dec_off"][i + 1]]), "long KAT %d (data_bits %d)" % (i, nb)
    # all of them in one call as well (several blocks share a wave in the packed kernels): group by size
    for nb in sorted(set(int(x) for x in g["data_bits"])):
        idx = [i for i, x in enumerate(g["data_bits"]) if int(x) == nb]
        s = np.concatenate([g["symbols"][g["sym_off"][i]:g["sym_off"][i + 1]] for i in idx])
        got = rx.conv_decode(s, nb, len(idx))
        for k, i in enumerate(idx):
            assert np.array_equal(got[k], g["decoded"][g["dec_off"][i]:g["dec_off"][i + 1]]), (nb, k)
    _set_viterbi(rx, VITERBI_KINDS[0])


@pytest.mark.parametrize("kind", VITERBI_KINDS, ids=_kind_id)
def test_conv_decode_random_vs_oracle(rx, po, kind):
    _set_viterbi(rx, kind)
    rng = np.random.default_rng(21)
    for nb in (1, 2, 7, 10, 58, 64, 91, 122, 130, 1000, 1001, 8418, 32826):
        nblk = 5 if nb < 5000 else 2
        n = 2 * (nb + 6)
        s = rng.integers(0, 256, nblk * n, dtype=np.uint8)
        d = rng.integers(0, 256, (nb + 13) // 8 + 1, dtype=np.uint8)
        e = np.clip(po.conv_encode(d, nb).astype(float) * 255 + rng.normal(0, 80, n), 0, 255).astype(np.uint8)
        e[2::6] = 127
        s[:n] = e
        got = rx.conv_decode(s, nb, nblk)
        for b in range(nblk):
            assert np.array_equal(got[b], po.conv_decode(s[b * n:(b + 1) * n], nb)), (nb, b)
    _set_viterbi(rx, VITERBI_KINDS[0])


# chain-back segmentations of the production kernels for the saturation / renormalisation corner cases below
SEGMENTATIONS = [(2, S, L) for S in (96, 960, 3072) for L in (0, 96)]


def _real_sse_decoder(po):
    """The REAL reference decoder (oracle/_ref, built from /root/reference in the dev container; the prebuilt library
    travels to the GPU box) if it is there, else None."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(po.__file__)), "_ref", "libfun_ofdm_ref.so")
    return po.Ref.conv_decode if os.path.exists(path) else None


@pytest.mark.parametrize("kind", SEGMENTATIONS, ids=_kind_id)
def test_conv_decode_saturation_corner_cases(rx, po, kind):
    """Where the uint8 saturation, the state-0 renormalisation rule and the tie rule decide the output:
    constant soft bytes (0, 255, 127 = erasures everywhere, 128), uniformly random bytes, alternating extremes, a clean
    codeword and a codeword with every second pair erased, at 18 (SIGNAL), 8418 (config 2) and 32826 (config 3) data bits."""
    _set_viterbi(rx, kind)
    rng = np.random.default_rng(77)
    for nb in (18, 8418, 32826):
        n = 2 * (nb + 6)
        d = rng.integers(0, 256, (nb + 13) // 8 + 1, dtype=np.uint8)
        code = po.conv_encode(d, nb).astype(np.uint8) * 255
        half = code.copy()
        half.reshape(-1, 2)[1::2] = 127
        ramp = (np.arange(n) * 7 % 256).astype(np.uint8)
        alt = np.where(np.arange(n) % 2 == 0, 0, 255).astype(np.uint8)
        alt4 = np.where((np.arange(n) // 2) % 2 == 0, 0, 255).astype(np.uint8)
        blocks = [np.full(n, v, np.uint8) for v in (0, 255, 127, 128, 1, 254)] + [rng.integers(0, 256, n, dtype=np.uint8) for _ in range(3)] + \
                 [alt, alt4, ramp, code, half, 255 - code]
        s = np.concatenate(blocks)
        got = rx.conv_decode(s, nb, len(blocks))
        for b, blk in enumerate(blocks):
            assert np.array_equal(got[b], po.conv_decode(blk, nb)), (nb, b)
    _set_viterbi(rx, VITERBI_KINDS[0])


def _corner_blocks(po, nb, rng):
    n = 2 * (nb + 6)
    d = rng.integers(0, 256, (nb + 13) // 8 + 1, dtype=np.uint8)
    code = po.conv_encode(d, nb).astype(np.uint8) * 255
    half = code.copy()
    half.reshape(-1, 2)[1::2] = 127
    ramp = (np.arange(n) * 7 % 256).astype(np.uint8)
    alt = np.where(np.arange(n) % 2 == 0, 0, 255).astype(np.uint8)
    return [np.full(n, v, np.uint8) for v in (0, 255, 127, 128, 1, 254)] + [rng.integers(0, 256, n, dtype=np.uint8) for _ in range(3)] + [alt, ramp, code, half, 255 - code]


def test_conv_decode_corner_cases_against_the_compiled_reference_decoder(rx, po):
    """The same corner cases against the REAL reference decoder (src/viterbi.cpp compiled in place into oracle/_ref): an explicit test of its
    own, so that a box on which the git-ignored oracle/_ref did not travel REPORTS a skip instead of silently asserting less.
    The committed golden vectors of that decoder (tests/golden/viterbi_long_ref.npz, test_conv_decode_matches_reference_sse_vectors) cover the same shapes
    wherever this one skips."""
    real = _real_sse_decoder(po)
    if real is None:
        pytest.skip("oracle/_ref/libfun_ofdm_ref.so is not on this box (it is built from /root/reference in the dev container and git-ignored); "
                    "the committed viterbi_long_ref.npz vectors of the same decoder are checked by test_conv_decode_matches_reference_sse_vectors")
    rng = np.random.default_rng(78)
    for kind in [(2, 960, 96), (2, 96, 0)]:
        _set_viterbi(rx, kind)
        for nb in (18, 8418, 32826):
            blocks = _corner_blocks(po, nb, rng)
            got = rx.conv_decode(np.concatenate(blocks), nb, len(blocks))
            for b, blk in enumerate(blocks):
                assert np.array_equal(got[b], real(blk, nb)), ("vs the compiled reference", kind, nb, b)
    _set_viterbi(rx, VITERBI_KINDS[0])


def test_fft_forward_vs_oracle(rx, po):
    rng = np.random.default_rng(22)
    v = rng.normal(size=(37, 64)) + 1j * rng.normal(size=(37, 64))
    got = rx.fft_forward(v)
    for i in range(v.shape[0]):
        want = po.fft64(v[i])
        assert _rel(got[i], want) < 1e-13
        assert _rel(got[i], np.fft.fftshift(np.fft.fft(v[i]))) < 1e-13


def test_golden_frames(rx, po, golden):
    g = golden.frames
    rx.set_option("record_eq", 1)
    soft_mismatch = soft_total = 0
    for name in g["names"]:
        iq, descs = g[name + "_iq"], g[name + "_desc"]
        psdu, res = rx.decode_frames_host(iq, descs, _ends(descs, iq.size))
        want = g[name + "_res"]
        assert [res[0]["status"], res[0]["rate"], res[0]["length"], res[0]["num_symbols"]] == list(want), name
        if want[0] == 0:
            assert np.array_equal(psdu[0, :want[2]], g[name + "_psdu"]), name
        if name + "_hinv" in g:
            t = rx.taps(1, eq=True)
            used = np.abs(po.lts_freq_domain()) > 0          # null carriers hold 0/NaN in the reference
            assert _rel(t["hinv"][0][used], g[name + "_hinv"][used].astype(np.complex128)) < REL_TOL, name
            eq = t["eq"][:t["eq_off"][1]].reshape(-1, 48)
            ref = g[name + "_eq"].astype(np.complex128).reshape(-1, 48)
            assert eq.shape == ref.shape, name
            for k in range(eq.shape[0]):
                assert _rel(eq[k], ref[k]) < REL_TOL, (name, k)
            soft = t["soft"][:t["soft_off"][1]]
            ws = g[name + "_soft"]
            assert soft.size == ws.size, name
            soft_mismatch += int((soft != ws).sum())
            soft_total += ws.size
    rx.set_option("record_eq", 0)
    # fp64 on both sides: soft bytes are expected to agree exactly
    assert soft_mismatch == 0, "%d of %d soft bytes differ" % (soft_mismatch, soft_total)


def test_golden_frames_with_the_real_reference_on_the_transmit_side(rx, po, golden):
    """tests/golden/frames_reftx.npz: coded bits out of the REAL reference's conv_encode / puncture / interleave for all eleven rates at 1
    and 4095 payload bytes (oracle/gen_golden.py); the device must hand back the PAYLOAD -- an expected output no restatement computed --
    with the status, rate and length the frame was built with, and agree with the oracle's receiver on every field."""
    from test_oracle_golden import _reftx_cases
    n = 0
    for name, rate, pay, s in _reftx_cases(po, golden):
        descs = po.find_alignments_f32(s)
        assert descs.size == 1, name
        psdu, res = rx.decode_frames_host(s, descs, _ends(descs, s.size))
        assert (res[0]["status"], res[0]["rate"], res[0]["length"]) == (0, rate, pay.size), (name, res[0])
        assert np.array_equal(psdu[0, :pay.size], pay), name
        ores, opsdu = po.decode_alignment_f32(s, descs[0])
        assert res[0]["num_symbols"] == ores["num_symbols"] and np.array_equal(psdu[0, :pay.size], opsdu[:pay.size]), name
        n += 1
    assert n == 22


def _make_stream(po, rng, specs, snr_db=25.0, gap=(150, 600), cfo_hz=0.0):
    parts, pays = [], []
    for rate, ln in specs:
        pay = rng.integers(0, 256, ln, dtype=np.uint8)
        f = po.build_frame(pay, rate)
        if cfo_hz:
            f = f * np.exp(2j * np.pi * rng.uniform(-cfo_hz, cfo_hz) * np.arange(f.size) / 20e6)
        f = f * np.exp(1j * rng.uniform(0, 2 * np.pi))
        parts += [np.zeros(int(rng.integers(*gap)), complex), f]
        pays.append(pay)
    parts.append(np.zeros(400, complex))
    s = np.concatenate(parts)
    sigma = np.sqrt(0.0124 / (2 * 10 ** (snr_db / 10)))
    s = s + (rng.normal(size=s.size) + 1j * rng.normal(size=s.size)) * sigma
    return s.astype(np.complex64), pays


@pytest.mark.parametrize("kind", VITERBI_KINDS, ids=_kind_id)
def test_mixed_rate_stream_vs_oracle(rx, po, kind):
    """All 11 rates, several lengths, marginal SNR so that some frames fail their CRC: identical
    status, header fields and PSDU bytes per frame; soft bytes identical."""
    _set_viterbi(rx, kind)
    rng = np.random.default_rng(23)
    specs = [(r, int(rng.integers(1, 400))) for r in range(11)] * 2 + [(10, 1024), (0, 37), (2, 1500), (9, 4095), (8, 1)]
    iq, pays = _make_stream(po, rng, specs, snr_db=19.0, cfo_hz=3000.0)
    descs = po.find_alignments_f32(iq)
    assert descs.size == len(specs)
    ends = _ends(descs, iq.size)
    psdu, res = rx.decode_frames_host(iq, descs, ends)
    t = rx.taps(descs.size)
    opsdu, ores = po.decode_batch_f32(iq, descs, ends, threads=4)
    n_ok = n_fail = 0
    for f in range(descs.size):
        assert tuple(res[f]) == tuple(ores[f]), (f, res[f], ores[f])
        if res[f]["status"] == 0:
            n_ok += 1
            assert np.array_equal(psdu[f, :res[f]["length"]], opsdu[f, :res[f]["length"]]), f
            assert np.array_equal(psdu[f, :res[f]["length"]], pays[f]), f
        else:
            n_fail += 1
        _, _, taps = po.decode_alignment_f32(iq, descs[f], end=ends[f], taps=True)
        if res[f]["rate"] >= 0 and res[f]["status"] in (0, 2):
            got = t["soft"][t["soft_off"][f]:t["soft_off"][f + 1]]
            assert np.array_equal(got, taps["soft"]), f
    assert n_ok >= 10 and n_fail >= 1, (n_ok, n_fail)      # the case really has both outcomes


def test_frame_pairs_of_extreme_length_mismatch(rx, po):
    """The forward pass packs frames 2i and 2i+1 into one wave.  Shortest against longest in one wave, in both orders, next to an
    odd frame out and a header failure: the short frame's blocks beyond its end (stored as ones), the long frame's last partial
    chunk and the late store of each chunk's third block are all in play.  Status, fields, PSDUs and every soft byte vs the oracle."""
    rx.set_option("record_soft", 1)
    _set_viterbi(rx, VITERBI_KINDS[0])
    rng = np.random.default_rng(97)
    specs = [(10, 1), (0, 4095), (0, 4095), (10, 0), (2, 3000), (10, 2), (9, 47), (0, 1300), (5, 4095), (8, 7), (3, 100)]
    iq, pays = _make_stream(po, rng, specs, snr_db=24.0, gap=(120, 300))
    descs = po.find_alignments_f32(iq)
    assert descs.size == len(specs)
    ends = _ends(descs, iq.size)
    # a garbage SIGNAL in the middle: frame 5 loses its header, its pair partner must not notice
    bad = int(descs["lts1_pos"][5]) + 144
    iq2 = iq.copy()
    iq2[bad:bad + 64] = (rng.normal(size=64) + 1j * rng.normal(size=64)).astype(np.complex64) * 0.05
    for stream in (iq, iq2):
        psdu, res = rx.decode_frames_host(stream, descs, ends)
        t = rx.taps(descs.size)
        opsdu, ores = po.decode_batch_f32(stream, descs, ends, threads=4)
        for f in range(descs.size):
            assert tuple(res[f]) == tuple(ores[f]), (f, res[f], ores[f])
            assert np.array_equal(psdu[f], opsdu[f]), f
            if res[f]["rate"] >= 0 and res[f]["status"] in (0, 2):
                _, _, taps = po.decode_alignment_f32(stream, descs[f], end=ends[f], taps=True)
                assert np.array_equal(t["soft"][t["soft_off"][f]:t["soft_off"][f + 1]], taps["soft"]), f
    assert (res["status"] == 0).sum() >= len(specs) - 2


def test_frontend_vs_oracle(rx, po):
    """The front end (k_header + k_data_symbols_q4) on a mixed-rate stream with CFO: status, PSDUs, every soft byte and the equalised
    carriers (1e-4 relative, north_star) against the oracle."""
    rx.set_option("record_soft", 1)
    rx.set_option("record_eq", 1)
    _set_viterbi(rx, VITERBI_KINDS[0])
    try:
        rng = np.random.default_rng(29)
        specs = [(r, int(rng.integers(20, 300))) for r in range(11)] + [(10, 1024), (5, 700), (0, 61), (8, 1)]
        iq, pays = _make_stream(po, rng, specs, snr_db=22.0, cfo_hz=2500.0)
        descs = po.find_alignments_f32(iq)
        assert descs.size == len(specs)
        ends = _ends(descs, iq.size)
        psdu, res = rx.decode_frames_host(iq, descs, ends)
        t = rx.taps(descs.size, eq=True)
        opsdu, ores = po.decode_batch_f32(iq, descs, ends, threads=4)
        for f in range(descs.size):
            assert tuple(res[f]) == tuple(ores[f]), (f, res[f], ores[f])
            if res[f]["status"] == 0:
                assert np.array_equal(psdu[f, :res[f]["length"]], pays[f]), f
            _, _, taps = po.decode_alignment_f32(iq, descs[f], end=ends[f], taps=True)
            if res[f]["rate"] < 0:
                continue
            got = t["soft"][t["soft_off"][f]:t["soft_off"][f + 1]]
            assert np.array_equal(got, taps["soft"]), f
            eq = t["eq"][t["eq_off"][f]:t["eq_off"][f + 1]].reshape(-1, 48)
            want = taps["eq"].reshape(-1, 48)
            assert eq.shape == want.shape
            for k in range(eq.shape[0]):
                assert _rel(eq[k], want[k]) < REL_TOL, (f, k)
    finally:
        rx.set_option("record_eq", 0)


def test_truncated_and_degenerate_inputs(rx, po, golden):
    g = golden.frames
    iq, descs = g["rate10_iq"], g["rate10_desc"]
    lts1 = int(descs[0]["lts1_pos"])
    # cut inside the data symbols, inside SIGNAL, and before the LTS is complete
    for cut, want_rate in ((lts1 + 1000, 10), (lts1 + 200, -1), (lts1 + 100, -1)):
        psdu, res = rx.decode_frames_host(iq, descs, np.array([cut], np.int64))
        o_res, _ = po.decode_alignment_f32(iq, descs[0], end=cut)
        assert res[0]["status"] == 3 == o_res["status"]
        assert res[0]["rate"] == want_rate
    # zero frames is a no-op
    psdu, res = rx.decode_frames_host(iq, descs[:0], np.zeros(0, np.int64))
    assert psdu.shape[0] == 0
    # pure noise "frame": both sides must agree (normally header failure)
    rng = np.random.default_rng(24)
    noise = (rng.normal(size=4000) + 1j * rng.normal(size=4000)).astype(np.complex64) * 0.05
    d = np.zeros(1, descs.dtype)
    d["lts1_pos"] = 100; d["rot_start"] = 90; d["c"] = 1.0; d["c_prev"] = 1.0
    psdu, res = rx.decode_frames_host(noise, d, np.array([4000], np.int64))
    o_res, _ = po.decode_alignment_f32(noise, d[0], end=4000)
    assert tuple(res[0]) == tuple(o_res)
    # all-zero samples: the channel estimate is NaN everywhere; the reference yields a header failure
    z = np.zeros(2000, np.complex64)
    psdu, res = rx.decode_frames_host(z, d, np.array([2000], np.int64))
    o_res, _ = po.decode_alignment_f32(z, d[0], end=2000)
    assert tuple(res[0]) == tuple(o_res)


def test_overlapping_alignments_exhaust_the_workspace(po):
    """The workspace of a call is sized from its sample count.  Descriptors that claim the same samples several times over
    need more than that: the frames that still fit decode as usual, the rest come back FOA_ST_NO_SPACE (with their symbol
    count), nothing is written out of bounds, and the handle works normally afterwards."""
    import fun_ofdm_amd as foa
    rng = np.random.default_rng(12)
    r = foa.Receiver(0)                       # a fresh handle: its buffers are exactly as large as this call needs
    try:
        for pipeline in (1, 0):
            r.set_option("pipeline", pipeline)
            iq, pays = _make_stream(po, rng, [(0, 100)], snr_db=30.0, gap=(150, 151))
            d1 = po.find_alignments_f32(iq)
            assert d1.size == 1
            for copies in (2, 3, 7, 300):
                descs = np.repeat(d1, copies)
                ends = np.full(copies, iq.size, np.int64)
                psdu, res = r.decode_frames_host(iq, descs, ends)
                st = res["status"]
                assert st[0] == foa.ST_OK and psdu[0, :100].tobytes() == pays[0].tobytes()
                fit = int((st == foa.ST_OK).sum())
                assert 1 <= fit < copies and (st[:fit] == foa.ST_OK).all() and (st[fit:] == foa.ST_NO_SPACE).all()
                assert all(psdu[k, :100].tobytes() == pays[0].tobytes() for k in range(fit))
                assert (res["num_symbols"] == res["num_symbols"][0]).all() and (res["rate"] == 0).all() and (res["length"] == 100).all()
            # and a normal call on the same handle
            iq2, pays2 = _make_stream(po, rng, [(10, 300), (5, 77)], snr_db=30.0)
            d2 = po.find_alignments_f32(iq2)
            psdu2, res2 = r.decode_frames_host(iq2, d2, _ends(d2, iq2.size))
            assert (res2["status"] == foa.ST_OK).all()
            assert psdu2[0, :300].tobytes() == pays2[0].tobytes() and psdu2[1, :77].tobytes() == pays2[1].tobytes()
    finally:
        r.close()


def test_rotation_switch_inside_lts(rx, po, golden):
    """rot_start after lts1_pos: the first samples of the LTS window use the previous phasor
    (timing_sync.cpp:105,124: the tag may sit up to 8 samples before the STS_END sample)."""
    g = golden.frames
    iq, descs = g["rate5_iq"], g["rate5_desc"].copy()
    descs["rot_start"] = descs["lts1_pos"] + 5
    descs["c_prev"], descs["s_prev"] = np.cos(0.3), np.sin(0.3)
    ends = np.array([iq.size], np.int64)
    psdu, res = rx.decode_frames_host(iq, descs, ends)
    o_res, o_psdu, taps = po.decode_alignment_f32(iq, descs[0], taps=True)
    assert tuple(res[0]) == tuple(o_res)
    t = rx.taps(1)
    used = np.abs(po.lts_freq_domain()) > 0
    assert _rel(t["hinv"][0][used], taps["hinv"][used]) < 1e-9
    assert np.array_equal(t["soft"][:t["soft_off"][1]], taps["soft"])


@pytest.mark.parametrize("kind", VITERBI_KINDS, ids=_kind_id)
def test_config2_shape_batch(rx, po, kind):
    """BASELINE config 2 at reduced count: 54 Mbps, 1024-byte payloads, 25 dB, frame pitch 4096;
    device-resident buffers through the device-pointer entry point."""
    import torch
    import fun_ofdm_amd as foa
    _set_viterbi(rx, kind)
    rng = np.random.default_rng(25)
    n_frames, pitch = 96, 4096
    iq = np.zeros(n_frames * pitch, np.complex64)
    pays = []
    sigma = np.sqrt(0.0124 / (2 * 10 ** 2.5))
    for f in range(n_frames):
        pay = rng.integers(0, 256, 1024, dtype=np.uint8)
        fr = po.build_frame(pay, 10) * np.exp(1j * rng.uniform(0, 2 * np.pi))
        assert fr.size == 3520
        iq[f * pitch + 176:f * pitch + 176 + 3520] = fr
        pays.append(pay)
    iq = (iq + (rng.normal(size=iq.size) + 1j * rng.normal(size=iq.size)) * sigma).astype(np.complex64)
    # timing_sync also fires on noise now and then (its correlation is normalised by power, not by
    # amplitude: timing_sync.cpp:79), so there can be more alignments than frames; all are decoded.
    descs = po.find_alignments_f32(iq)
    assert descs.size >= n_frames
    m = descs.size
    ends = _ends(descs, iq.size)
    dev = torch.device("cuda", 0)
    t_iq = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).to(dev)
    t_desc = torch.from_numpy(descs.view(np.uint8).copy()).to(dev)
    t_ends = torch.from_numpy(ends).to(dev)
    t_psdu = torch.zeros((m, 1024), dtype=torch.uint8, device=dev)
    t_res = torch.zeros((m, 4), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    rx.decode_frames_dev(t_iq, t_desc, t_ends, t_psdu, t_res)
    rx.sync()
    res = t_res.cpu().numpy()
    psdu = t_psdu.cpu().numpy()
    real = np.nonzero((descs["lts1_pos"] - 360) % pitch == 0)[0]      # frame start 176 + LTS1 at 184
    assert real.size == n_frames
    assert (res[real, 0] == 0).all() and (res[real, 1] == 10).all() and (res[real, 2] == 1024).all() and (res[real, 3] == 39).all()
    for k, f in enumerate(real):
        assert np.array_equal(psdu[f], pays[k]), f
    opsdu, ores = po.decode_batch_f32(iq, descs, ends, slot_bytes=1024, threads=4)
    assert np.array_equal(ores.view(np.int32).reshape(-1, 4), res)
    ok = res[:, 0] == 0
    assert np.array_equal(opsdu[ok], psdu[ok])
    ms = rx.kernel_ms()
    assert ms["total"] > 0 and ms["viterbi_fwd"] > 0


@pytest.mark.parametrize("rate", [0, 2, 3, 5, 6, 8, 9, 10])
def test_config3_rate_sweep_4096_byte_psdu(rx, po, rate):
    """BASELINE config 3: the eight 802.11a rates with a 4096-byte PSDU (4092-byte payload + CRC-32; the 12-bit
    length field cannot express more): identical status / PSDU as the oracle, frame by frame."""
    from fun_ofdm_amd import synth
    import fun_ofdm_amd as foa
    _set_viterbi(rx, VITERBI_KINDS[0])
    n = 3
    pays = synth.splitmix64_bytes(0x0FD3 + rate, n, 4092)
    frames = synth.build_frames(pays, rate)
    pitch = ((frames.shape[1] + 700) // 4096 + 1) * 4096
    iq, _ = synth.make_stream(frames, pitch, 200, 25.0, seed=300 + rate)
    descs = foa.find_alignments(iq)
    assert descs.tobytes() == po.find_alignments_f32(iq).tobytes()
    ends = foa.alignment_ends(descs, iq.size)
    psdu, res = rx.decode_frames_host(iq, descs, ends)
    opsdu, ores = po.decode_batch_f32(iq, descs, ends, threads=4)
    assert np.array_equal(res.view(np.int32), ores.view(np.int32))
    ok = res["status"] == 0
    assert ok.sum() >= n - 1                      # 9 Mbps long frames fail now and then in the reference too
    assert np.array_equal(psdu[ok], opsdu[ok])
    real = np.nonzero((descs["lts1_pos"] - 384) % pitch == 0)[0]
    for k, f in enumerate(real):
        if res[f]["status"] == 0:
            assert np.array_equal(psdu[f, :4092], pays[k])


def test_config5_back_to_back_mixed_rates_with_cfo(rx, po):
    """BASELINE config 5 shape: one continuous stream, frames of the 8 rates back to back (zero gap), per-frame CFO
    within +-4 kHz (the reference has no CFO estimator), 25 dB."""
    from fun_ofdm_amd import synth
    import fun_ofdm_amd as foa
    rng = np.random.default_rng(55)
    parts, pays = [np.zeros(300, complex)], []
    for i in range(24):
        rate = (0, 2, 3, 5, 6, 8, 9, 10)[i % 8]
        pay = synth.splitmix64_bytes(900 + i, 1, 1024)[0]
        f = synth.build_frames(pay[None, :], rate)[0]
        f = f * np.exp(2j * np.pi * rng.uniform(-4000, 4000) * np.arange(f.size) / 20e6 + 1j * rng.uniform(0, 6.28))
        parts.append(f)
        pays.append(pay)
    parts.append(np.zeros(600, complex))
    s = np.concatenate(parts)
    s = (s + (rng.normal(size=s.size) + 1j * rng.normal(size=s.size)) * np.sqrt(0.0124 / 2 / 10 ** 2.5)).astype(np.complex64)
    descs = foa.find_alignments(s)
    ends = foa.alignment_ends(descs, s.size)
    psdu, res = rx.decode_frames_host(s, descs, ends)
    opsdu, ores = po.decode_batch_f32(s, descs, ends, threads=4)
    assert np.array_equal(res.view(np.int32), ores.view(np.int32))
    ok = res["status"] == 0
    assert np.array_equal(psdu[ok], opsdu[ok])
    got = [psdu[f, :1024].tobytes() for f in np.nonzero(ok)[0]]
    chain = po.ReceiverChain().run_stream(s.astype(np.complex128))
    assert got == chain                           # same ordered PSDU list as the reference-shaped chain
    assert len(got) >= 20


def test_device_sync_matches_host_sync(rx, po):
    """frame_detector + timing_sync on the device against the host restatement (which equals the reference bit for bit):
    same alignments (LTS1 position, rotation start) and the same phasors to 1e-12; then PSDUs decoded from the
    device-made descriptors equal the oracle's."""
    import torch
    import fun_ofdm_amd as foa
    from fun_ofdm_amd import synth
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(77)
    for trial, (snr, cfo, gap) in enumerate(((25.0, None, 0), (20.0, 3000.0, 1), (25.0, 4000.0, 2))):
        parts, pays = [], []
        for i in range(40):
            rate = (0, 2, 3, 5, 6, 8, 9, 10)[i % 8]
            pay = synth.splitmix64_bytes(500 + 100 * trial + i, 1, 64 + 13 * i)[0]
            f = synth.build_frames(pay[None, :], rate)[0] * np.exp(1j * rng.uniform(0, 6.28))
            if cfo:
                f = f * np.exp(2j * np.pi * rng.uniform(-cfo, cfo) * np.arange(f.size) / 20e6)
            parts += [np.zeros(int(rng.integers(0, 900)) if gap else 0, complex), f]
            pays.append(pay)
        parts.append(np.zeros(777, complex))
        s = np.concatenate([np.zeros(123, complex)] + parts)
        s = (s + (rng.normal(size=s.size) + 1j * rng.normal(size=s.size)) * np.sqrt(0.0124 / 2 / 10 ** (snr / 10))).astype(np.complex64)
        want = foa.find_alignments(s)
        t_iq = torch.from_numpy(s.view(np.float32).reshape(-1, 2)).to(dev)
        cap = s.size // 300 + 16
        t_desc = torch.zeros(cap * 48, dtype=torch.uint8, device=dev)
        t_ends = torch.zeros(cap, dtype=torch.int64, device=dev)
        n = rx.sync_dev(t_iq, t_desc, t_ends)
        got = t_desc.cpu().numpy()[:n * 48].view(foa.frame_desc_dtype)
        assert n == want.size, (trial, n, want.size)
        assert np.array_equal(got["lts1_pos"], want["lts1_pos"]) and np.array_equal(got["rot_start"], want["rot_start"])
        for k in ("c", "s", "c_prev", "s_prev"):
            assert np.abs(got[k] - want[k]).max() < 1e-12, k
        ends = t_ends.cpu().numpy()[:n]
        assert np.array_equal(ends, foa.alignment_ends(want, s.size))
        # decode from the device-made descriptors without leaving the device
        t_psdu = torch.zeros((n, 4096), dtype=torch.uint8, device=dev)
        t_res = torch.zeros((n, 4), dtype=torch.int32, device=dev)
        rx.decode_frames_dev(t_iq, t_desc[:n * 48], t_ends[:n], t_psdu, t_res)
        rx.sync()
        opsdu, ores = po.decode_batch_f32(s, want, ends, threads=4)
        res = t_res.cpu().numpy()
        assert np.array_equal(res, ores.view(np.int32).reshape(-1, 4))
        ok = res[:, 0] == 0
        assert np.array_equal(t_psdu.cpu().numpy()[ok], opsdu[ok]) and ok.sum() >= 30


def test_device_sync_at_the_detection_threshold(rx, po):
    """Adversarial for the device pre-sync's `equal up to ties` claim: frames at 8-12 dB, where frame_detector's
    normalised lag-16 correlation and timing_sync's LTS correlation hover around their 0.9 thresholds for hundreds of samples per
    frame (the host restatement sums with the reference's running accumulators, the device forms every window directly).  The
    descriptors must still be equal -- a decision could only differ within ~1e-15 of the threshold -- and the test checks that the
    stream really sits there: thousands of windows within 1e-2 of 0.9, dozens within 1e-4, and detections on both sides."""
    import torch
    import fun_ofdm_amd as foa
    from fun_ofdm_amd import synth
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(4242)
    parts, n_frames = [np.zeros(300, complex)], 240
    for i in range(n_frames):
        pay = synth.splitmix64_bytes(9000 + i, 1, 40 + (i % 7) * 9)[0]
        f = synth.build_frames(pay[None, :], (0, 3, 6, 10)[i % 4])[0] * np.exp(1j * rng.uniform(0, 6.28)) * 10 ** rng.uniform(-0.5, 1.2)
        snr = rng.uniform(8.0, 12.0)
        sigma = np.sqrt(np.mean(np.abs(f[:320]) ** 2) / (2 * 10 ** (snr / 10)))
        seg = np.concatenate([f, np.zeros(int(rng.integers(40, 400)), complex)])
        parts.append(seg + (rng.normal(size=seg.size) + 1j * rng.normal(size=seg.size)) * sigma)
    s = np.concatenate(parts).astype(np.complex64)
    # how close to the threshold the stream runs (plain numpy, windows formed directly)
    x = s.astype(np.complex128)
    prod = x[16:] * np.conj(x[:-16])
    pw = np.abs(x[16:]) ** 2
    cs = np.concatenate([[0], np.cumsum(prod)])
    ps = np.concatenate([[0], np.cumsum(pw)])
    c = np.abs(cs[16:] - cs[:-16]) / np.maximum(ps[16:] - ps[:-16], 1e-300)
    assert np.count_nonzero(np.abs(c - 0.9) < 1e-2) > 2000 and np.count_nonzero(np.abs(c - 0.9) < 1e-4) > 20
    want = foa.find_alignments(s)
    assert 20 < want.size < n_frames            # some frames are found, some are not: the thresholds are really in play
    t_iq = torch.from_numpy(s.view(np.float32).reshape(-1, 2)).to(dev)
    cap = s.size // 300 + 64
    t_desc = torch.zeros(cap * 48, dtype=torch.uint8, device=dev)
    t_ends = torch.zeros(cap, dtype=torch.int64, device=dev)
    n = rx.sync_dev(t_iq, t_desc, t_ends)
    got = t_desc.cpu().numpy()[:n * 48].view(foa.frame_desc_dtype)
    assert n == want.size, (n, want.size)
    assert np.array_equal(got["lts1_pos"], want["lts1_pos"]) and np.array_equal(got["rot_start"], want["rot_start"])
    for k in ("c", "s", "c_prev", "s_prev"):
        assert np.abs(got[k] - want[k]).max() < 1e-12, k
    # ... and the host restatement is the oracle's (= the compiled reference's, tests/test_oracle_vs_ref.py)
    assert np.array_equal(po.find_alignments_f32(s)["lts1_pos"], want["lts1_pos"])


def test_device_sync_in_two_halves_pipelined_with_decode(rx, po):
    """foa_rx_sync_dev_begin / _end: (a) the same descriptors, ends and count as the blocking call; (b) used the way it is meant --
    _end(k), _begin(k+1), decode(k) over a series of DIFFERENT streams with two descriptor sets -- every batch's PSDUs equal the
    oracle's from the host-made descriptors; (c) the state errors: a second _begin, an _end without _begin, and the degenerate
    empty stream."""
    import torch
    import fun_ofdm_amd as foa
    from fun_ofdm_amd import synth
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(31337)
    streams = []
    for b in range(7):
        parts = [np.zeros(int(rng.integers(50, 400)), complex)]
        for i in range(24 + 3 * b):
            pay = synth.splitmix64_bytes(70000 + 100 * b + i, 1, 30 + 41 * ((i + b) % 9))[0]
            f = synth.build_frames(pay[None, :], (0, 2, 3, 5, 6, 8, 9, 10)[(i + b) % 8])[0] * np.exp(1j * rng.uniform(0, 6.28))
            f = f * np.exp(2j * np.pi * rng.uniform(-3000, 3000) * np.arange(f.size) / 20e6)
            parts += [f, np.zeros(int(rng.integers(0, 300)), complex)]
        s = np.concatenate(parts)
        s = (s + (rng.normal(size=s.size) + 1j * rng.normal(size=s.size)) * np.sqrt(0.0124 / 2 / 10 ** 2.5)).astype(np.complex64)
        streams.append(s)
    t_iq = [torch.from_numpy(s.view(np.float32).reshape(-1, 2)).to(dev) for s in streams]
    cap = max(s.size for s in streams) // 300 + 16
    sets = [(torch.zeros(cap * 48, dtype=torch.uint8, device=dev), torch.zeros(cap, dtype=torch.int64, device=dev)) for _ in range(2)]
    # (a) against the blocking call
    b_desc, b_ends = torch.zeros_like(sets[0][0]), torch.zeros_like(sets[0][1])
    nb = rx.sync_dev(t_iq[0], b_desc, b_ends)
    rx.sync_dev_begin(t_iq[0], *sets[0])
    with pytest.raises(foa.FoaError):
        rx.sync_dev_begin(t_iq[0], *sets[1])               # one in flight per handle
    assert rx.sync_dev_end() == nb and nb >= 20
    assert torch.equal(sets[0][0][:nb * 48], b_desc[:nb * 48]) and torch.equal(sets[0][1][:nb], b_ends[:nb])
    with pytest.raises(foa.FoaError):
        rx.sync_dev_end()                                  # nothing in flight
    # (b) pipelined over different streams
    outs = [(torch.zeros((cap, 512), dtype=torch.uint8, device=dev), torch.zeros((cap, 4), dtype=torch.int32, device=dev)) for _ in streams]
    ns = []
    rx.sync_dev_begin(t_iq[0], *sets[0])
    for k in range(len(streams)):
        n = rx.sync_dev_end()
        if k + 1 < len(streams):
            rx.sync_dev_begin(t_iq[k + 1], *sets[(k + 1) % 2])
        d, e = sets[k % 2]
        rx.decode_frames_dev(t_iq[k], d[:n * 48], e[:n], outs[k][0][:n], outs[k][1][:n])
        ns.append(n)
    rx.sync()
    for k, s in enumerate(streams):
        want = foa.find_alignments(s)
        assert ns[k] == want.size, (k, ns[k], want.size)
        ends = foa.alignment_ends(want, s.size)
        opsdu, ores = po.decode_batch_f32(s, want, ends, slot_bytes=512, threads=4)
        r = outs[k][1][:ns[k]].cpu().numpy()
        assert np.array_equal(r, ores.view(np.int32).reshape(-1, 4)), k
        ok = r[:, 0] == 0
        assert ok.sum() >= 20 and np.array_equal(outs[k][0][:ns[k]].cpu().numpy()[ok], opsdu[ok]), k
    # (c) a capacity too small for the stream: the count is refused at _end (FOA_E_INVALID), nothing is written beyond the capacity, and the
    # handle takes the next pre-sync as if nothing had happened
    small_d = torch.full((5 * 48 + 64,), 0xA5, dtype=torch.uint8, device=dev)
    small_e = torch.full((5 + 8,), -7, dtype=torch.int64, device=dev)
    assert rx._lib.foa_rx_sync_dev_begin(rx._h, t_iq[0].data_ptr(), t_iq[0].shape[0], small_d.data_ptr(), small_e.data_ptr(), 5) == 0
    with pytest.raises(foa.FoaError):
        rx.sync_dev_end()
    assert (small_d[5 * 48:] == 0xA5).all() and (small_e[5:] == -7).all()
    rx.sync_dev_begin(t_iq[0], *sets[0])
    assert rx.sync_dev_end() == nb and torch.equal(sets[0][0][:nb * 48], b_desc[:nb * 48])
    # (d) an empty stream: nothing queued, _end reports 0
    assert rx._lib.foa_rx_sync_dev_begin(rx._h, t_iq[0].data_ptr(), 0, sets[0][0].data_ptr(), sets[0][1].data_ptr(), cap) == 0
    assert rx.sync_dev_end() == 0


def test_device_sync_edge_inputs(rx):
    """Streams too short to hold a window, all-zero input (0/0 everywhere: never above threshold) and noise only:
    the device stage agrees with the host restatement and writes nothing it should not."""
    import torch
    import fun_ofdm_amd as foa
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    cases = [np.zeros(5000, np.complex64)]
    for n in (1, 17, 100, 1023, 1024, 1025, 4097):
        cases.append(((rng.normal(size=n) + 1j * rng.normal(size=n)) * 0.05).astype(np.complex64))
    for s in cases:
        want = foa.find_alignments(s)
        t_iq = torch.from_numpy(s.view(np.float32).reshape(-1, 2)).to(dev)
        t_desc = torch.full((64 * 48,), 0xA5, dtype=torch.uint8, device=dev)
        t_ends = torch.full((64,), -7, dtype=torch.int64, device=dev)
        n = rx.sync_dev(t_iq, t_desc, t_ends)
        assert n == want.size, (s.size, n, want.size)
        got = t_desc.cpu().numpy()
        assert np.array_equal(got[:n * 48].view(foa.frame_desc_dtype)["lts1_pos"], want["lts1_pos"])
        assert (got[n * 48:] == 0xA5).all() and (t_ends.cpu().numpy()[n:] == -7).all()


def test_device_sync_decides_timing_sync_99_by_the_reference_call_size(rx, po):
    """The device pre-sync drops the alignment the reference drops at a boundary of its 4096-sample calls (`if(lts_offset < 0) break;`,
    timing_sync.cpp:99; tests/test_synth.py has the host side of this), keeps it when told to decide as one call (option
    "sync_call" 0), and a stream engine fed in chunks of any size drops the same frame."""
    import torch
    import fun_ofdm_amd as foa
    from fun_ofdm_amd import synth
    dev = torch.device("cuda", 0)
    pays = synth.splitmix64_bytes(31, 3, 100)
    iq, _ = synth.make_stream(synth.build_frames(pays, 5), pitch=3000, lead=500, snr_db=25.0, seed=8)
    base = foa.find_alignments(iq, call=0)
    x = int(base["rot_start"][1])
    rel = int(base["lts1_pos"][1]) - 24 + 32 - x
    assert base.size == 3 and 0 < rel < 32

    def device(stream):
        t_iq = torch.from_numpy(stream.view(np.float32).reshape(-1, 2)).to(dev)
        t_desc = torch.zeros(64 * 48, dtype=torch.uint8, device=dev)
        t_ends = torch.zeros(64, dtype=torch.int64, device=dev)
        n = rx.sync_dev(t_iq, t_desc, t_ends)
        return t_desc.cpu().numpy()[:n * 48].view(foa.frame_desc_dtype).copy()

    dropped_somewhere = False
    for late in range(0, 4):
        pad = (-(x + 160) + late) % 4096
        s = np.concatenate([np.zeros(pad, np.complex64), iq])
        want = po.find_alignments_f32(s)
        dropped = late + rel < 32
        dropped_somewhere |= dropped
        assert want.size == (2 if dropped else 3)
        got = device(s)
        assert got.size == want.size and np.array_equal(got["lts1_pos"], want["lts1_pos"]) and np.array_equal(got["rot_start"], want["rot_start"]), late
        rx.set_option("sync_call", 0)
        assert device(s).size == 3
        rx.set_option("sync_call", 4096)
        if dropped:                                      # the stream engine: absolute positions, not buffer positions, decide
            r = foa.Receiver(0)
            st = foa.Stream(r, 8192, 2)
            got_p, i, rng = [], 0, np.random.default_rng(late)
            while i < s.size:
                n = int(rng.integers(1, 5000))
                got_p += st.push(s[i:i + n])
                i += n
            got_p += st.flush()
            st.close(); r.close()
            assert got_p == [pays[0].tobytes(), pays[2].tobytes()]
    assert dropped_somewhere
    with pytest.raises(foa.FoaError):
        rx.set_option("sync_call", 100)


def test_device_sync_non_finite_samples(rx, po):
    """NaN and infinite samples.  A NaN costs the reference exactly the two products it is part of: circular_accumulator.h:88-95
    takes a NaN sample in as zero.  The device stage does the same (a window whose quick test is not finite is summed again term by
    term under that rule), so its descriptors equal the reference's with a NaN in a gap, in a short training sequence, in a long
    training sequence and in the payload.  An INFINITE sample is different: the reference's running sums turn NaN when it leaves
    the window and stay NaN for good -- frame_detector tags nothing behind it (the host restatement reproduces that) --, while the
    device stage sums every window from its own sixteen terms and finds the frames behind it: its descriptors equal the
    reference's on the stream with that one sample zeroed.  Same class of deliberate, documented difference as the glitch below."""
    import torch
    import fun_ofdm_amd as foa
    from fun_ofdm_amd import synth
    dev = torch.device("cuda", 0)
    pays = synth.splitmix64_bytes(78, 8, 150)
    iq, _ = synth.make_stream(synth.build_frames(pays, 5), 4096, 300, 25.0, seed=4)
    clean = foa.find_alignments(iq)
    assert clean.size == 8

    def device(stream):
        t_iq = torch.from_numpy(stream.view(np.float32).reshape(-1, 2)).to(dev)
        t_desc = torch.zeros(64 * 48, dtype=torch.uint8, device=dev)
        t_ends = torch.zeros(64, dtype=torch.int64, device=dev)
        n = rx.sync_dev(t_iq, t_desc, t_ends)
        return t_desc.cpu().numpy()[:n * 48].view(foa.frame_desc_dtype).copy(), t_iq, t_desc, t_ends, n

    def same(got, want):
        if got.size != want.size or not (np.array_equal(got["lts1_pos"], want["lts1_pos"]) and np.array_equal(got["rot_start"], want["rot_start"])):
            return False
        return all(np.abs(got[k] - want[k]).max() < 1e-12 for k in ("c", "s", "c_prev", "s_prev")) if got.size else True

    gap = 3 * 4096 + 3000                                # behind the fourth frame (150 bytes at 18 Mbps end well before)
    start = 4 * 4096 + 300                               # first sample of the fifth frame
    # NaN: the reference's rule, reproduced -- wherever it falls (gap, STS, the STS/LTS boundary, LTS, payload; one part or both)
    for pos in (gap, start + 40, start + 100, start + 150, start + 158, start + 161, start + 200, start + 290, start + 700):
        for bad in (complex(np.nan, 0.0), complex(0.3, np.nan), complex(np.nan, np.nan)):
            g = iq.copy()
            g[pos] = bad
            host = foa.find_alignments(g)
            assert host.tobytes() == po.find_alignments_f32(g).tobytes()
            got = device(g)[0]
            assert same(got, host), (pos, bad, got["lts1_pos"], host["lts1_pos"])
    assert foa.find_alignments(np.where(np.arange(iq.size) == gap, np.complex64(complex(np.nan, 0)), iq).astype(np.complex64)).size == 8
    # infinite: the reference goes blind, the device stage does not
    for bad in (complex(np.inf, 0.0), complex(-np.inf, np.nan), complex(1.0, -np.inf)):
        g = iq.copy()
        g[gap] = bad
        z = iq.copy()
        z[gap] = 0
        want = foa.find_alignments(z)
        assert want.tobytes() == clean.tobytes()         # (the zeroed sample changes no decision)
        host = foa.find_alignments(g)
        assert host.tobytes() == po.find_alignments_f32(g).tobytes()
        assert host.size == 4                            # nothing behind the sample
        got, t_iq, t_desc, t_ends, n = device(g)
        assert n == 8 and same(got, want)
        t_psdu = torch.zeros((n, 256), dtype=torch.uint8, device=dev)
        t_res = torch.zeros((n, 4), dtype=torch.int32, device=dev)
        rx.decode_frames_dev(t_iq, t_desc[:n * 48], t_ends[:n], t_psdu, t_res)
        rx.sync()
        assert (t_res.cpu().numpy()[:, 0] == 0).all() and np.array_equal(t_psdu.cpu().numpy()[:, :150], pays)


def test_device_sync_large_dynamic_range(rx, po):
    """One huge sample (an ADC glitch) in front of normal frames.  frame_detector's running sums (circular_accumulator.h:88-95:
    sum -= old; sum += new) never forget such a sample exactly -- after 1e10 or more the residue of the cancelled power term
    makes the reference tag spurious plateaus for the rest of the stream, on top of the real frames.  The device stage forms
    every 16-term window directly, so the glitch is gone once it has left the window: it finds the real frames and none of
    the residue's artefacts.  That difference is deliberate and documented (include/fun_ofdm_amd.h); a caller
    that needs the reference's decisions bit for bit on such input uses foa_sync_push_*, which equals the reference here
    too.  Up to a dynamic range of ~1e6 in amplitude the two agree exactly."""
    import torch
    import fun_ofdm_amd as foa
    from fun_ofdm_amd import synth
    dev = torch.device("cuda", 0)
    pays = synth.splitmix64_bytes(77, 6, 200)
    iq, _ = synth.make_stream(synth.build_frames(pays, 8), 4096, 300, 25.0, seed=3)
    clean = foa.find_alignments(iq)
    real = clean["lts1_pos"]
    assert real.size == 6

    def device(stream):
        t_iq = torch.from_numpy(stream.view(np.float32).reshape(-1, 2)).to(dev)
        cap = stream.size // 300 + 16
        t_desc = torch.zeros(cap * 48, dtype=torch.uint8, device=dev)
        t_ends = torch.zeros(cap, dtype=torch.int64, device=dev)
        n = rx.sync_dev(t_iq, t_desc, t_ends)
        return t_desc.cpu().numpy()[:n * 48].view(foa.frame_desc_dtype), t_iq, t_desc, t_ends, n

    for val in (30.0, 1e4, 1e10, 1e30):
        g = iq.copy()
        g[100] = val
        host = foa.find_alignments(g)
        assert host.tobytes() == po.find_alignments_f32(g).tobytes()          # the host restatement is the reference's behaviour
        got, t_iq, t_desc, t_ends, n = device(g)
        assert set(real) <= set(host["lts1_pos"])
        if val <= 1e4:
            assert np.array_equal(got["lts1_pos"], host["lts1_pos"]) and np.array_equal(got["rot_start"], host["rot_start"])
        else:
            assert host.size > real.size                                       # the reference's residue artefacts ...
            assert np.array_equal(got["lts1_pos"], real)                       # ... which the windowed sums do not have
        # and the frames decode from the device-made descriptors
        t_psdu = torch.zeros((n, 256), dtype=torch.uint8, device=dev)
        t_res = torch.zeros((n, 4), dtype=torch.int32, device=dev)
        rx.decode_frames_dev(t_iq, t_desc[:n * 48], t_ends[:n], t_psdu, t_res)
        rx.sync()
        res, psdu = t_res.cpu().numpy(), t_psdu.cpu().numpy()
        on = [int(np.nonzero(got["lts1_pos"] == p)[0][0]) for p in real]
        assert all(res[a, 0] == 0 and psdu[a, :200].tobytes() == pays[k].tobytes() for k, a in enumerate(on))


def test_device_sync_between_pipelined_decode_calls(rx, po):
    """sync k+1 is queued while decode k is still in flight (it runs on the third stream under that call's forward
    pass): descriptors, PSDUs and results of every round equal those of the same round run alone."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(91)
    streams = []
    for k in range(4):
        specs = [(int(rng.integers(0, 11)), int(rng.integers(1, 700))) for _ in range(8 + 2 * k)]
        iq, _ = _make_stream(po, rng, specs, snr_db=25.0)
        streams.append(iq)
    def run(piped):
        rx.set_option("pipeline", 1 if piped else 0)
        outs = []
        for iq in streams:
            t_iq = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).to(dev)
            cap = iq.size // 300 + 16
            t_desc = torch.zeros(cap * 48, dtype=torch.uint8, device=dev)
            t_ends = torch.zeros(cap, dtype=torch.int64, device=dev)
            n = rx.sync_dev(t_iq, t_desc, t_ends)
            t_psdu = torch.zeros((n, 4096), dtype=torch.uint8, device=dev)
            t_res = torch.zeros((n, 4), dtype=torch.int32, device=dev)
            rx.decode_frames_dev(t_iq, t_desc[:n * 48], t_ends[:n], t_psdu, t_res)
            if not piped:
                rx.sync()
            outs.append((t_iq, t_desc, t_ends, t_psdu, t_res, n))
        rx.sync()
        return [(o[5], o[1].cpu().numpy(), o[2].cpu().numpy(), o[3].cpu().numpy(), o[4].cpu().numpy()) for o in outs]
    alone, piped = run(False), run(True)
    rx.set_option("pipeline", 1)
    for a, b in zip(alone, piped):
        assert a[0] == b[0] and a[0] >= 6
        for x, y in zip(a[1:], b[1:]):
            assert np.array_equal(x, y)


def test_pipelined_calls_keep_their_results_apart(rx, po):
    """Back-to-back decode calls without a sync in between (the finish of call k runs on a second stream under the
    forward pass of call k+1, on alternating work sets): every call must produce exactly what it produces alone."""
    import torch
    dev = torch.device("cuda", 0)
    _set_viterbi(rx, VITERBI_KINDS[0])
    rng = np.random.default_rng(31)
    cases = []
    for k in range(5):
        specs = [(int(rng.integers(0, 11)), int(rng.integers(1, 600))) for _ in range(6 + 3 * k)]
        iq, pays = _make_stream(po, rng, specs, snr_db=24.0)
        descs = po.find_alignments_f32(iq)
        ends = _ends(descs, iq.size)
        cases.append((iq, descs, ends))
    rx.set_option("pipeline", 0)
    alone = [rx.decode_frames_host(iq, d, e) for iq, d, e in cases]
    rx.set_option("pipeline", 1)
    bufs = []
    for iq, d, e in cases:
        t_iq = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).to(dev)
        t_d = torch.from_numpy(d.view(np.uint8).copy()).to(dev)
        t_e = torch.from_numpy(e).to(dev)
        t_p = torch.zeros((d.size, 4096), dtype=torch.uint8, device=dev)
        t_r = torch.zeros((d.size, 4), dtype=torch.int32, device=dev)
        bufs.append((t_iq, t_d, t_e, t_p, t_r))
    torch.cuda.synchronize()
    for depth in (0, 2, 3, 4):                             # loops in flight: by grid size (these small batches: four), or fixed
        rx.set_option("depth", depth)
        for b in bufs:
            b[3].zero_(); b[4].zero_()
        torch.cuda.synchronize()
        for rep in range(3):                               # 15 calls in flight order, no sync between them
            for b in bufs:
                rx.decode_frames_dev(*b)
        rx.sync()
        for (psdu, res), b in zip(alone, bufs):
            assert np.array_equal(b[4].cpu().numpy(), res.view(np.int32).reshape(-1, 4)), depth
            assert np.array_equal(b[3].cpu().numpy(), psdu), depth
    rx.set_option("depth", 0)
    ms = rx.kernel_ms()
    assert ms["viterbi_finish"] > 0 and rx.kernel_ms(previous=True)["viterbi_fwd"] > 0
    # a host-pointer call and a pre-sync + decode round in the same arrangement (their copies and the pre-sync run off the lanes)
    iq, d, e = cases[-1]
    psdu, res = rx.decode_frames_host(iq, d, e)
    assert np.array_equal(res.view(np.int32), alone[-1][1].view(np.int32)) and np.array_equal(psdu, alone[-1][0])
    t_iq, t_d, t_e, t_p, t_r = bufs[-1]
    for _ in range(3):
        n = rx.sync_dev(t_iq, t_d, t_e)
        assert n == d.size
        rx.decode_frames_dev(t_iq, t_d[:n * 48], t_e[:n], t_p[:n], t_r[:n])
    rx.sync()
    assert np.array_equal(t_r.cpu().numpy(), alone[-1][1].view(np.int32).reshape(-1, 4)) and np.array_equal(t_p.cpu().numpy(), alone[-1][0])


@pytest.mark.parametrize("seed", [101, 102, 103])
def test_random_batches_vs_oracle(rx, po, seed):
    """Randomised batches (rates, lengths incl. 0 and 4095, SNR from hopeless to clean, CFO, gaps, a cut-off last frame),
    decoded with the defaults (pipelined calls, front end by context) and with short chain-back segments: every alignment
    must come out exactly as the oracle has it."""
    rng = np.random.default_rng(seed)
    for rep in range(3):
        n = int(rng.integers(20, 60))
        specs = []
        for _ in range(n):
            r = int(rng.integers(0, 11))
            ln = int(rng.choice([0, 1, 4095, int(rng.integers(2, 400)), int(rng.integers(400, 2000))], p=[0.05, 0.05, 0.03, 0.6, 0.27]))
            specs.append((r, ln))
        snr = float(rng.choice([6.0, 14.0, 20.0, 27.0]))
        iq, pays = _make_stream(po, rng, specs, snr_db=snr, gap=(0, 700), cfo_hz=float(rng.choice([0.0, 2000.0, 4500.0])))
        if rep == 2:
            iq = iq[:iq.size - int(rng.integers(500, 3000))]              # the stream ends inside the last frame
        descs = po.find_alignments_f32(iq)
        if descs.size == 0:
            continue
        ends = _ends(descs, iq.size)
        opsdu, ores = po.decode_batch_f32(iq, descs, ends, threads=4)
        for kind in (VITERBI_KINDS[0], (2, 96, 96), VITERBI_KINDS[1]):
            _set_viterbi(rx, kind)
            psdu, res = rx.decode_frames_host(iq, descs, ends)
            assert np.array_equal(res.view(np.int32), ores.view(np.int32)), (seed, rep, kind, snr)
            ok = res["status"] == 0
            assert np.array_equal(psdu[ok], opsdu[ok]), (seed, rep, kind)
    _set_viterbi(rx, VITERBI_KINDS[0])


def test_large_mixed_batch_vs_oracle(rx, po):
    """More than kSingleBelow - 1 (2 048) alignments in ONE call, so that the forward pass takes two frames per wave (smaller calls take one): random
    rates and lengths side by side in a wave's two halves (different step counts, a dead alignment next to a live one, an odd count), some
    frames in noise.  Every alignment exactly as the oracle has it; the same stream cut into small calls (one frame per wave) must agree too."""
    rng = np.random.default_rng(77)
    specs = [(int(rng.integers(0, 11)), int(rng.choice([0, 1, int(rng.integers(2, 120)), int(rng.integers(120, 500))], p=[0.03, 0.03, 0.7, 0.24]))) for _ in range(2900)]
    iq, pays = _make_stream(po, rng, specs, snr_db=17.0, gap=(0, 300), cfo_hz=2500.0)
    descs = po.find_alignments_f32(iq)
    assert descs.size >= 2200                                            # (> 2 048: two frames per wave)
    if descs.size % 2 == 0:
        descs = descs[:-1]                                               # an odd count: the last wave has one frame
    ends = _ends(descs, iq.size)
    opsdu, ores = po.decode_batch_f32(iq, descs, ends, threads=8)
    assert 1400 < int(np.count_nonzero(ores["status"] == 0)) and int(np.count_nonzero(ores["status"] != 0)) > 20
    psdu, res = rx.decode_frames_host(iq, descs, ends)
    assert np.array_equal(res.view(np.int32), ores.view(np.int32))
    ok = res["status"] == 0
    assert np.array_equal(psdu[ok], opsdu[ok])
    for lo, n2 in [(lo, 500) for lo in range(0, descs.size, 500)] + [(0, 1500), (600, 1501)]:   # ... and in calls of 500 and 1 500 alignments (one frame per wave: one / two workgroups per CU)
        d2, e2 = descs[lo:lo + n2], ends[lo:lo + n2].copy()
        e2[-1] = min(int(e2[-1]), iq.size)
        p2, r2 = rx.decode_frames_host(iq, d2, e2)
        o2p, o2r = po.decode_batch_f32(iq, d2, e2, threads=8)
        assert np.array_equal(r2.view(np.int32), o2r.view(np.int32)), lo
        ok2 = r2["status"] == 0
        assert np.array_equal(p2[ok2], o2p[ok2]), lo


@pytest.mark.parametrize("pipeline", [1, 0])
def test_async_host_calls(rx, po, pipeline):
    """foa_rx_submit_host / foa_rx_collect: several calls in flight, the caller's buffers reusable at once, results
    identical to the synchronous entry point and delivered in submission order."""
    import time
    _set_viterbi(rx, VITERBI_KINDS[0])
    rng = np.random.default_rng(41)
    cases = []
    for k in range(6):
        specs = [(int(rng.integers(0, 11)), int(rng.integers(1, 500))) for _ in range(4 + 2 * k)]
        iq, pays = _make_stream(po, rng, specs, snr_db=24.0)
        descs = po.find_alignments_f32(iq)
        cases.append((iq, descs, _ends(descs, iq.size)))
    rx.set_option("pipeline", 0)
    want = [rx.decode_frames_host(iq, d, e) for iq, d, e in cases]
    rx.set_option("pipeline", pipeline)
    try:
        tickets = []
        for iq, d, e in cases:
            iq2, d2, e2 = iq.copy(), d.copy(), e.copy()
            tickets.append(rx.submit_host(iq2, d2, e2))
            iq2[:] = 0; d2["lts1_pos"] = -1; e2[:] = 0            # the library has its own copy by now
        # poll the first without blocking until it is there, then take the rest in order
        t0 = time.time()
        first = None
        while first is None and time.time() - t0 < 30:
            first = rx.collect(tickets[0], wait=False)
        assert first is not None
        got = [first] + [rx.collect(t) for t in tickets[1:]]
        for (psdu, res), (wp, wr) in zip(got, want):
            assert np.array_equal(res.view(np.int32), wr.view(np.int32))
            assert np.array_equal(psdu, wp)
        # more than 8 in flight is refused, and collecting makes room again
        many = [rx.submit_host(*cases[0]) for _ in range(8)]
        with pytest.raises(Exception):
            rx.submit_host(*cases[0])
        for t in many:
            psdu, res = rx.collect(t)
            assert np.array_equal(psdu, want[0][0])
    finally:
        rx.set_option("pipeline", 1)


def test_context_alignments_and_explicit_ends_follow_the_restatement(rx, po):
    """foa_rx_submit_host_ctx / foa_rx_decode_frames_ctx_dev: a stream decoded in two pieces, the rest handed over as CONTEXT, gives the
    results of the one-piece decode (a frame cut short by a later LTS1 fills on with the context's vectors); without context such a frame is
    FOA_ST_TRUNCATED; ends that are not the next alignment's LTS1 make alignments independent.  Against fo_decode_batch_v2_f32."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "manual"))
    import stress_tags
    import fun_ofdm_amd as foa
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(8)
    n_split = n_trunc = n_late = 0
    for seed in range(60):
        s, d = stress_tags.make_case(seed)
        if d.size < 3:
            continue
        ends = np.append(d["lts1_pos"][1:], s.size).astype(np.int64)
        k = int(rng.integers(1, d.size))
        close = np.nonzero(np.diff(d["lts1_pos"]) < 64)[0]
        if close.size and seed % 2:                                    # every other case with a pile-up: cut it in the middle
            k = int(close[0]) + 1
        # (a) host entry with context
        want_p, want_r = po.decode_batch_v2_f32(s, d, ends, n_ctx=d.size - k)
        got_p, got_r = rx.collect(rx.submit_host(s, d, ends, n_context=d.size - k))
        assert np.array_equal(got_r.view(np.int32), want_r.view(np.int32)) and np.array_equal(got_p, want_p), (seed, k)
        # (b) device entry with context
        t_iq = torch.from_numpy(s.view(np.float32).reshape(-1, 2)).to(dev)
        t_d = torch.from_numpy(d.view(np.uint8).copy()).to(dev)
        t_e = torch.from_numpy(ends).to(dev)
        t_p = torch.zeros((k, 4096), dtype=torch.uint8, device=dev)
        t_r = torch.zeros((k, 4), dtype=torch.int32, device=dev)
        rx.decode_frames_dev(t_iq, t_d, t_e, t_p, t_r, n_context=d.size - k)
        rx.sync()
        assert np.array_equal(t_r.cpu().numpy(), want_r.view(np.int32).reshape(-1, 4)) and np.array_equal(t_p.cpu().numpy(), want_p), (seed, k)
        # (c) the first piece alone: what needed the rest is TRUNCATED, nothing else changes
        p0, r0 = rx.decode_frames_host(s, d[:k], ends[:k])
        o0p, o0r = po.decode_batch_v2_f32(s, d[:k], ends[:k])
        assert np.array_equal(r0.view(np.int32), o0r.view(np.int32)) and np.array_equal(p0, o0p), (seed, k)
        differ = r0["status"] != want_r["status"]
        assert np.all(r0["status"][differ] == foa.ST_TRUNCATED)
        n_split += 1
        n_trunc += int(np.count_nonzero(differ))
        # (d) unlinked ends
        e2 = ends - rng.integers(1, 40, d.size)
        p2, r2 = rx.decode_frames_host(s, d, e2)
        o2p, o2r = po.decode_batch_v2_f32(s, d, e2)
        assert np.array_equal(r2.view(np.int32), o2r.view(np.int32)) and np.array_equal(p2, o2p), (seed, "unlinked")
        assert not np.any(r2["status"] == foa.ST_SUPERSEDED)
        # (e) the second piece with the first handed over as LEAD (foa_rx_decode_frames_lead_ctx_dev): the results of the one-piece decode --
        # an alignment less than 64 samples behind the last alignment of the first piece is read one symbol late, as there; without the lead it
        # is decoded as if nothing were in front of it
        full_p, full_r = po.decode_batch_v2_f32(s, d, ends)
        m2 = d.size - k
        t_p2 = torch.zeros((m2, 4096), dtype=torch.uint8, device=dev)
        t_r2 = torch.zeros((m2, 4), dtype=torch.int32, device=dev)
        rx.decode_frames_dev(t_iq, t_d, t_e, t_p2, t_r2, n_lead=k)
        rx.sync()
        assert np.array_equal(t_r2.cpu().numpy(), full_r[k:].view(np.int32).reshape(-1, 4)) and np.array_equal(t_p2.cpu().numpy(), full_p[k:]), (seed, k, "lead")
        if d["lts1_pos"][k] - d["lts1_pos"][k - 1] < 64:
            n_late += 1
            alone_p, alone_r = rx.decode_frames_host(s, d[k:], ends[k:])
            o_p, o_r = po.decode_batch_v2_f32(s, d[k:], ends[k:])
            assert np.array_equal(alone_r.view(np.int32), o_r.view(np.int32)) and np.array_equal(alone_p, o_p), (seed, k, "no lead")
    assert n_split > 40 and n_trunc > 5 and n_late >= 3, (n_split, n_trunc, n_late)


def test_many_frames_of_one_to_three_symbols_and_groups_of_mixed_frames(rx, po):
    """The data-symbol kernel stages the channel estimates of a wave's sixteen symbols through LDS, four alignments' worth; a group that
    holds more alignments than that -- frames of one to three symbols back to back, as here -- reads its taps from memory, and groups that
    straddle frames of different rates take the generic demapper.  Every such path against the oracle, soft bytes included."""
    rng = np.random.default_rng(4242)
    specs = []
    for _ in range(260):
        rate = int(rng.choice((10, 9, 8, 6, 5, 3, 0)))
        # payloads that give 1 .. 3 symbols at the frame's rate (ppdu.cpp:40-44), now and then a long one in between
        dbps = po.rate_params(rate)["dbps"]
        nsym = int(rng.integers(1, 4))
        ln = max(0, (nsym * dbps - 22) // 8 - 4 - int(rng.integers(0, 2)))
        specs.append((rate, ln if rng.random() > 0.05 else int(rng.integers(200, 700))))
    s, pays = _make_stream(po, rng, specs, snr_db=27.0, gap=(40, 90))
    descs = po.find_alignments_f32(s)
    ends = _ends(descs, s.size)
    rx.set_option("record_soft", 1)
    psdu, res = rx.decode_frames_host(s, descs, ends)
    opsdu, ores = po.decode_batch_f32(s, descs, ends)
    assert np.array_equal(res.view(np.int32).reshape(-1, 4), ores.view(np.int32).reshape(-1, 4))
    ok = res["status"] == 0
    assert ok.sum() >= 200 and np.array_equal(psdu[ok], opsdu[ok])
    short = (res["num_symbols"][ok] <= 3).sum()
    assert short >= 150, short
    t = rx.taps(descs.size)
    for f in np.nonzero(ok)[0][:60]:
        _, _, ot = po.decode_alignment_f32(s, descs[f], end=int(ends[f]), taps=True)
        assert np.array_equal(t["soft"][t["soft_off"][f]:t["soft_off"][f + 1]], ot["soft"]), f


def test_work_sets_sized_by_max_dbps_and_the_promise_is_checked_on_the_device(po):
    """Option max_dbps: results are identical for every value that covers the capture's rates; a frame of a higher rate than promised, once
    the work set is full, is FOA_ST_NO_SPACE -- reported, never decoded wrongly -- and the frames that fit are untouched."""
    import fun_ofdm_amd as foa
    rng = np.random.default_rng(77)
    s, pays = _make_stream(po, rng, [(0, 300)] * 6 + [(2, 200)] * 4, snr_db=28.0)
    descs = po.find_alignments_f32(s)
    ends = _ends(descs, s.size)
    opsdu, ores = po.decode_batch_f32(s, descs, ends)
    for dbps in (216, 48, 36):
        r = foa.Receiver(0)
        r.set_option("max_dbps", dbps)
        psdu, res = r.decode_frames_host(s, descs, ends)
        assert np.array_equal(res.view(np.int32).reshape(-1, 4), ores.view(np.int32).reshape(-1, 4)), dbps
        assert np.array_equal(psdu[res["status"] == 0], opsdu[ores["status"] == 0]), dbps
        r.close()
    # a dense capture of 54 Mbps frames under a promise of 6 Mbps: some fit into the slack of the work set, the rest are told so
    s2, pays2 = _make_stream(po, rng, [(10, 1500)] * 40, snr_db=28.0, gap=(60, 80))
    d2 = po.find_alignments_f32(s2)
    e2 = _ends(d2, s2.size)
    o2p, o2r = po.decode_batch_f32(s2, d2, e2)
    r = foa.Receiver(0)
    r.set_option("max_dbps", 24)
    psdu, res = r.decode_frames_host(s2, d2, e2)
    st = res["status"]
    assert (st == foa.ST_NO_SPACE).sum() >= 5, st
    fits = st != foa.ST_NO_SPACE
    assert np.array_equal(res[fits].view(np.int32).reshape(-1, 4), o2r[fits].view(np.int32).reshape(-1, 4))
    assert np.array_equal(psdu[st == 0], o2p[st == 0])
    assert np.array_equal(res["rate"], o2r["rate"]) and np.array_equal(res["length"], o2r["length"])      # (the header fields are reported either way)
    for bad in (23, 217, 0):
        with pytest.raises(foa.FoaError):
            r.set_option("max_dbps", bad)
    r.close()
