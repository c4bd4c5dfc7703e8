"""Host logic: the numpy frame generator and the C-ABI library surface (CPU only)."""
import ctypes
import os
import re

import numpy as np

import fun_ofdm_amd as foa
from fun_ofdm_amd import synth


def test_build_frames_matches_oracle(po):
    rng = np.random.default_rng(31)
    for rate in range(11):
        for length in (0, 1, 57, 100 + rate):
            pays = rng.integers(0, 256, (3, length), dtype=np.uint8)
            got = synth.build_frames(pays, rate)
            assert got.shape[1] == synth.frame_samples(rate, length) == po.frame_samples(rate, length)
            for i in range(3):
                want = po.build_frame(pays[i], rate)
                assert np.abs(got[i] - want).max() < 1e-15, (rate, length)
    assert np.array_equal(synth.preamble(), po.preamble_samples())


def test_splitmix_payloads_are_deterministic_and_distinct():
    a = synth.splitmix64_bytes(0x0FD2, 50, 1024)
    b = synth.splitmix64_bytes(0x0FD2, 50, 1024)
    assert np.array_equal(a, b) and len({x.tobytes() for x in a}) == 50
    assert abs(a.mean() - 127.5) < 2


def test_stream_decodes_in_oracle_chain(po):
    pays = synth.splitmix64_bytes(7, 6, 300)
    frames = synth.build_frames(pays, 10)
    iq, starts = synth.make_stream(frames, pitch=4096, lead=176, snr_db=25.0, seed=3)
    assert iq.dtype == np.complex64 and iq.size == 6 * 4096
    out = po.ReceiverChain().run_stream(iq.astype(np.complex128))
    assert out == [p.tobytes() for p in pays]


def test_host_sync_equals_oracle_sync(po, golden):
    """foa_sync_* (product, host side) makes the same decisions as the oracle's frame_detector +
    timing_sync, for any chunking."""
    s = golden.blocks_ref["stream"]
    want = po.find_alignments_f32(s)
    assert foa.find_alignments(s).tobytes() == want.tobytes() and want.size == 2
    pays = synth.splitmix64_bytes(9, 40, 64)
    iq, _ = synth.make_stream(synth.build_frames(pays, 3), pitch=2048, lead=200, snr_db=22.0, seed=5, cfo_hz=4000.0)
    want = po.find_alignments_f32(iq)
    assert want.size >= 38
    rng = np.random.default_rng(6)
    sy, got, x = foa.Sync(), [], 0
    while x < iq.size:
        n = int(rng.integers(1, 5000))
        got.append(sy.push(iq[x:x + n]))
        x += n
    got.append(sy.push(np.zeros(400, np.complex64)))
    got = np.concatenate(got)
    assert got[got["lts1_pos"] < iq.size].tobytes() == want.tobytes()
    # double-precision input path
    assert foa.find_alignments(iq.astype(np.complex128)).tobytes() == want.tobytes()


def _same_alignments(a, b):
    return a.size == b.size and np.array_equal(a["lts1_pos"], b["lts1_pos"]) and np.array_equal(a["rot_start"], b["rot_start"]) and \
        (a.size == 0 or max(np.abs(a[k] - b[k]).max() for k in ("c", "s", "c_prev", "s_prev")) < 1e-12)


def test_host_sync_decides_timing_sync_99_by_the_reference_call_size(po):
    """`if(lts_offset < 0) break;` (timing_sync.cpp:99) is relative to the working buffer of the call that walks over an STS_END: the
    reference drops an alignment whose LTS guard interval would start before it -- with the STS_END tag a sample or two late (the usual
    case), a frame whose STS_END is the first sample a call walks over, i.e. lies 160 samples before a call boundary.  The reference's
    receiver makes its calls 4096 samples long (receiver.h:16), and so does the oracle's one-shot.  The host restatement must drop the
    same frame, HOWEVER it is pushed; told to decide as one call (call = 0) it keeps it."""
    pays = synth.splitmix64_bytes(31, 3, 100)
    frames = synth.build_frames(pays, 5)
    iq, _ = synth.make_stream(frames, pitch=3000, lead=500, snr_db=25.0, seed=8)
    base = foa.find_alignments(iq, call=0)
    assert base.size == 3
    x = int(base["rot_start"][1])                        # STS_END sample of the middle frame
    rel = int(base["lts1_pos"][1]) - 24 + 32 - x         # its strongest LTS peak, relative to x
    assert 0 < rel < 32                                  # the tag is late: the guard interval starts before x
    hits = 0
    for late in range(0, 4):                             # x lands on input index 0, 1, 2, 3 of a call's buffer
        pad = (-(x + 160) + late) % 4096
        s = np.concatenate([np.zeros(pad, np.complex64), iq])
        want = po.find_alignments_f32(s)                 # the reference's blocks, 4096 samples per call
        dropped = late + rel < 32
        hits += dropped
        assert want.size == (2 if dropped else 3), (late, rel, want.size)
        assert _same_alignments(foa.find_alignments(s), want), late
        assert foa.find_alignments(s, call=0).size == 3
        rng = np.random.default_rng(late)
        sy, got, i = foa.Sync(), [], 0
        while i < s.size:                                # pushed in pieces that have nothing to do with 4096
            n = int(rng.integers(1, 7000))
            got.append(sy.push(s[i:i + n]))
            i += n
        got.append(sy.push(np.zeros(4096, np.complex64)))
        got = np.concatenate(got)
        assert _same_alignments(got[got["lts1_pos"] < s.size], want), late
        # another call size moves the boundary: 4000-sample calls keep the frame here, and drop it where THEIR boundary falls
        assert foa.find_alignments(s, call=4000).size == 3 or (x + pad + 160 - late) % 4000 < 4
    assert hits >= 1


def test_library_exports_every_declared_symbol():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # the product header (what INTEGRATION.md binds) and the diagnostics header (same library, no stability promise)
    hdr = open(os.path.join(root, "include", "fun_ofdm_amd.h")).read()
    diag = open(os.path.join(root, "include", "fun_ofdm_amd_diag.h")).read()
    product = set(re.findall(r"\b(foa_[a-z0-9_]+)\s*\(", hdr))
    lab = set(re.findall(r"\b(foa_[a-z0-9_]+)\s*\(", diag)) - {"foa_rx_set_option"}
    assert len(product) >= 20 and not (product & lab)
    # the lab stays out of the product header: timings, taps, decisions and the probe are diagnostics
    for name in ("foa_rx_probe_issue", "foa_rx_probe_h2d", "foa_rx_get_taps", "foa_rx_get_decisions", "foa_rx_forward_spacing_ms", "foa_rx_last_kernel_ms"):
        assert name in lab and name not in product, name
    assert "record_eq" not in hdr and "record_soft" not in hdr
    declared = product | lab
    L = ctypes.CDLL(foa.library_path())
    for name in sorted(declared):
        assert hasattr(L, name), name
    from fun_ofdm_amd._lib import EXPORTS
    assert declared == set(EXPORTS)
    assert L.foa_version() == 110


def test_no_cpu_fallback_without_gpu():
    """Without a HIP device the product fails loudly instead of computing on the CPU."""
    if foa.lib().foa_device_count() > 0:
        return
    try:
        foa.Receiver(0)
    except foa.FoaError as e:
        assert "no HIP device" in str(e)
    else:
        raise AssertionError("Receiver() must not succeed without a GPU")
