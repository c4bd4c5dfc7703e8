"""receiver_chain::process_samples() entirely on the device (foa_stream_*, fun_ofdm_amd/csrc/stream_engine.h): the stream is cut
into overlapping batches, every batch is pre-synchronised and decoded by the kernels of the batch path.  The ordered payload
list must equal the reference-shaped chain's (oracle) whatever the batch size and however the pushes are cut -- frames that
straddle one or several batch boundaries, back-to-back frames, long silences and a second preamble inside a frame included."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stream(po, rng, specs, snr_db=25.0, gap=(0, 700), cfo_hz=0.0):
    parts, pays = [np.zeros(200, complex)], []
    for rate, ln in specs:
        pay = rng.integers(0, 256, ln, dtype=np.uint8)
        f = po.build_frame(pay, rate)
        if cfo_hz:
            f = f * np.exp(2j * np.pi * rng.uniform(-cfo_hz, cfo_hz) * np.arange(f.size) / 20e6)
        f = f * np.exp(1j * rng.uniform(0, 2 * np.pi))
        parts += [f, np.zeros(int(rng.integers(*gap)), complex)]
        pays.append(pay.tobytes())
    parts.append(np.zeros(600, complex))
    s = np.concatenate(parts)
    sigma = np.sqrt(0.0124 / (2 * 10 ** (snr_db / 10)))
    return (s + (rng.normal(size=s.size) + 1j * rng.normal(size=s.size)) * sigma).astype(np.complex64), pays


@pytest.fixture(scope="module")
def rx():
    import fun_ofdm_amd as foa
    r = foa.Receiver(0)
    yield r
    r.close()


@pytest.fixture(scope="module")
def mixed(po):
    rng = np.random.default_rng(61)
    specs = [(int(rng.integers(0, 11)), int(rng.integers(1, 900))) for _ in range(40)]
    specs += [(0, 4095), (10, 1024), (2, 3000), (10, 0), (5, 1), (0, 2047), (9, 4095)]      # 109 840-sample frames: longer than most batches
    iq, pays = _stream(po, rng, specs, snr_db=22.0, cfo_hz=3000.0)
    want = po.ReceiverChain().run_stream(iq.astype(np.complex128))
    assert len(want) >= 40
    return iq, pays, want


@pytest.mark.parametrize("batch,chunk", [(4096, 4096), (4096, 1000), (20000, 4096), (65536, 7777), (1 << 18, 4096), (1 << 22, 100000)])
def test_stream_engine_equals_reference_chain(rx, mixed, batch, chunk):
    import fun_ofdm_amd as foa
    iq, pays, want = mixed
    st = foa.Stream(rx, batch)
    got = []
    try:
        for a in range(0, iq.size, chunk):
            got += st.push(iq[a:a + chunk])
        got += st.flush()
        stats = st.stats()
    finally:
        st.close()
    assert got == want, (batch, chunk, len(got), len(want))
    assert stats["samples"] == iq.size and stats["ok"] == len(want) and stats["batches"] == iq.size // batch + 1


def test_stream_engine_double_input_and_helper_threads(rx, mixed):
    """foa_stream_push_f64 (what process_samples hands over) with the narrowing spread over helper threads."""
    import fun_ofdm_amd as foa
    iq, pays, want = mixed
    wide = iq.astype(np.complex128)
    st = foa.Stream(rx, 1 << 17, narrow_threads=3)
    try:
        got = st.push(wide[:50000]) + st.push(wide[50000:50001]) + st.push(wide[50001:]) + st.flush()
    finally:
        st.close()
    assert got == want


def test_stream_engine_back_to_back_and_silence(rx, po):
    """Zero-gap frames (config 5 shape) across many small batches, then a silence longer than several batches, then more."""
    import fun_ofdm_amd as foa
    rng = np.random.default_rng(62)
    a, _ = _stream(po, rng, [((0, 2, 3, 5, 6, 8, 9, 10)[i % 8], 1024) for i in range(16)], gap=(0, 1), cfo_hz=4000.0)
    b, _ = _stream(po, rng, [(10, 300), (0, 50)], gap=(0, 300))
    sigma = np.sqrt(0.0124 / (2 * 10 ** 2.5))
    quiet = ((rng.normal(size=300000) + 1j * rng.normal(size=300000)) * sigma).astype(np.complex64)
    iq = np.concatenate([a, quiet, b])
    want = po.ReceiverChain().run_stream(iq.astype(np.complex128))
    assert len(want) >= 16
    for batch in (8192, 50000):
        st = foa.Stream(rx, batch)
        try:
            got = st.push(iq) + st.flush()
        finally:
            st.close()
        assert got == want, batch


def test_stream_engine_second_preamble_inside_a_frame(rx, po):
    from test_gpu_cpp_adaptors import _collision_stream
    import fun_ofdm_amd as foa
    for case in ("valid", "invalid", "none"):
        iq, pays = _collision_stream(po, case, 5)
        want = po.ReceiverChain().run_stream(iq.astype(np.complex128))
        for batch in (4096, 1 << 20):
            st = foa.Stream(rx, batch)
            try:
                got = st.push(iq) + st.flush()
            finally:
                st.close()
            assert got == want, (case, batch)


def test_stream_engine_api_edges(rx):
    import fun_ofdm_amd as foa
    with pytest.raises(foa.FoaError):
        foa.Stream(rx, 100)                                 # batch too small
    st = foa.Stream(rx, 4096)
    try:
        assert st.take() == [] and st.flush() == []         # an empty stream delivers nothing
        with pytest.raises(foa.FoaError):
            st.push(np.zeros(10, np.complex64))             # no pushes after the flush
    finally:
        st.close()
    # the handle is usable as before once the stream is closed
    psdu, res = rx.decode_frames_host(np.zeros(5000, np.complex64), np.zeros(0, foa.frame_desc_dtype), np.zeros(0, np.int64))
    assert psdu.shape[0] == 0


def test_process_samples_device_mode_through_the_cpp_chain(tmp_path, po, mixed):
    """fun_amd::receiver / receiver_chain with device_batch_samples > 0 (examples/foa_sim --device-batch): the same ordered
    payloads as the reference chain, from an fc32 and an fc64 capture, with and without the in-memory preload loop."""
    import fun_ofdm_amd as foa
    iq, pays, want = mixed
    exe = str(tmp_path / "foa_sim")
    libdir = os.path.dirname(foa.library_path())
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "examples", "foa_sim.cpp"), "-I", os.path.join(ROOT, "include"),
                    "-L", libdir, "-lfun_ofdm_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", exe], check=True)
    for fmt, data, extra in (("fc32", iq, ["--device-batch", "30000"]), ("fc64", iq.astype(np.complex128), ["--device-batch", "262144", "--narrow-threads", "2"]),
                             ("fc32", iq, ["--device-batch", "100000", "--preload", "--chunk", "65536", "--narrow-threads", "2"])):
        src, out = str(tmp_path / ("cap." + fmt)), str(tmp_path / ("psdus." + fmt))
        data.tofile(src)
        r = subprocess.run([exe, src, "--format", fmt, "--out", out] + extra, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        raw, recs, o = open(out, "rb").read(), [], 0
        while o < len(raw):
            n = int.from_bytes(raw[o:o + 4], "little")
            recs.append(raw[o + 4:o + 4 + n])
            o += 4 + n
        assert recs == want, (fmt, extra, len(recs), len(want))


def test_one_stream_per_handle_and_handle_destroyed_first(po):
    """a second foa_stream_create on a handle whose stream is open is refused (FOA_E_STATE) instead of corrupting the
    first one's state; destroying the HANDLE while its stream is open stops the engine (its threads use the handle), after which every
    call on the stream fails cleanly and destroying the stream only frees the shell."""
    import fun_ofdm_amd as foa
    from fun_ofdm_amd._lib import FoaError
    rng = np.random.default_rng(5)
    s, pays = _stream(po, rng, [(10, 300), (0, 120), (8, 500)])
    r = foa.Receiver(0)
    st = foa.Stream(r, 8192, 2)
    with pytest.raises(FoaError):
        foa.Stream(r, 8192, 0)
    import torch
    t_iq = torch.zeros((4096, 2), dtype=torch.float32, device="cuda:0")
    with pytest.raises(FoaError):                      # the engine owns the handle's pre-sync scratch too
        r.sync_dev(t_iq, torch.zeros(64 * 48, dtype=torch.uint8, device="cuda:0"), torch.zeros(64, dtype=torch.int64, device="cuda:0"))
    got = st.push(s) + st.flush()
    assert got == pays
    st.close()
    st2 = foa.Stream(r, 16384, 1)                   # ... and the handle takes a new stream once the first is gone
    got = st2.push(s[:4000])
    r.close()                                          # the handle goes first
    with pytest.raises(FoaError):
        st2.push(s[4000:])
    st2.close()


# ---- the same engine over several devices (foa_shard_*).  A test box with one GPU names it once, twice, three times (handles sharing it):
# what is tested then is the dealing of batches, the host-side carries and the chain state through the handles.  On a node with more GPUs the
# lists below widen by themselves: every prefix [0 .. n-1] of the devices there are (torch.cuda.device_count() does not start the runtime).
def _device_prefixes():
    try:
        import torch
        n = torch.cuda.device_count()
    except Exception:
        n = 0
    return [list(range(k)) for k in range(2, n + 1)]


_SHARD_CASES = [([0], 4096, 4096), ([0, 0], 4096, 1000), ([0, 0], 20000, 4096), ([0, 0, 0], 65536, 7777), ([0, 0], 1 << 18, 4096), ([0, 0], 1 << 22, 100000)] + \
               [(d, 16384, 4096) for d in _device_prefixes()] + [(d, 1 << 18, 65536) for d in _device_prefixes()]


def test_multi_device_cases_cover_the_devices_of_this_box():
    """The multi-device tests enumerate the box: with N >= 2 devices every list [0 .. n-1], n = 2 .. N, is among the cases."""
    import fun_ofdm_amd as foa
    n = foa.lib().foa_device_count()
    assert n >= 1
    real = [d for d, _, _ in _SHARD_CASES if len(set(d)) == len(d) and len(d) > 1]
    assert sorted(set(len(d) for d in real)) == list(range(2, n + 1))


@pytest.mark.parametrize("devices,batch,chunk", _SHARD_CASES)
def test_shard_engine_equals_reference_chain(mixed, devices, batch, chunk):
    import fun_ofdm_amd as foa
    iq, pays, want = mixed
    sh = foa.Shard(devices, batch, narrow_threads=2)
    got = []
    try:
        for a in range(0, iq.size, chunk):
            got += sh.push(iq[a:a + chunk])
        got += sh.flush()
        stats = sh.stats()
    finally:
        sh.close()
    assert got == want, (devices, batch, chunk, len(got), len(want))
    assert stats["samples"] == iq.size and stats["ok"] == len(want) and stats["batches"] == iq.size // batch + 1
    assert sum(stats["per_device_alignments"]) == stats["alignments"]
    if len(devices) > 1 and stats["batches"] > 4:
        assert all(v > 0 for v in stats["per_device_alignments"])          # every handle did part of the work


def test_shard_engine_phasor_chain_across_devices(po):
    """Long silences between frames with CFO: the first frame after a silence longer than the carry finds no alignment before it in its own
    buffer, so the phasor timing_sync had left in force comes from a batch that ANOTHER handle decoded (shard_core.h)."""
    import fun_ofdm_amd as foa
    rng = np.random.default_rng(63)
    sigma = np.sqrt(0.0124 / (2 * 10 ** 2.5))
    parts = []
    for i in range(6):
        a, _ = _stream(po, rng, [(int(rng.choice((0, 5, 10))), int(rng.integers(20, 600))) for _ in range(2)], gap=(0, 200), cfo_hz=4000.0)
        parts += [a, ((rng.normal(size=130000) + 1j * rng.normal(size=130000)) * sigma).astype(np.complex64)]
    iq = np.concatenate(parts)
    want = po.ReceiverChain().run_stream(iq.astype(np.complex128))
    assert len(want) >= 10
    for devices, batch in [([0, 0], 32768), ([0, 0, 0], 16384)] + [(d, 16384) for d in _device_prefixes()]:
        sh = foa.Shard(devices, batch)
        try:
            got = sh.push(iq) + sh.flush()
        finally:
            sh.close()
        assert got == want, (devices, batch)


def test_shard_engine_second_preamble_inside_a_frame_and_api_edges(po):
    from test_gpu_cpp_adaptors import _collision_stream
    import fun_ofdm_amd as foa
    for case in ("valid", "invalid"):
        iq, pays = _collision_stream(po, case, 5)
        want = po.ReceiverChain().run_stream(iq.astype(np.complex128))
        sh = foa.Shard([0, 0], 4096)
        try:
            got = sh.push(iq) + sh.flush()
            with pytest.raises(foa.FoaError):
                sh.push(np.zeros(10, np.complex64))             # no pushes after the flush
        finally:
            sh.close()
        assert got == want, case
    with pytest.raises(foa.FoaError):
        foa.Shard([0], 100)                                     # batch too small
    with pytest.raises(foa.FoaError):
        foa.Shard([], 4096)                                     # no device
    with pytest.raises(foa.FoaError):
        foa.Shard([0, 99], 4096)                                # no such device


def test_process_samples_over_a_device_list_through_the_cpp_chain(tmp_path, po, mixed):
    """fun_amd::receiver_chain(std::vector<int> devices, ...) behind examples/foa_sim --devices: the same ordered payloads."""
    import fun_ofdm_amd as foa
    iq, pays, want = mixed
    exe = str(tmp_path / "foa_sim")
    libdir = os.path.dirname(foa.library_path())
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "examples", "foa_sim.cpp"), "-I", os.path.join(ROOT, "include"),
                    "-L", libdir, "-lfun_ofdm_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", exe], check=True)
    src, out = str(tmp_path / "cap.fc32"), str(tmp_path / "psdus")
    iq.tofile(src)
    for extra in (["--devices", "0,0", "--device-batch", "30000"], ["--devices", "0,0,0", "--device-batch", "65536", "--narrow-threads", "2", "--preload", "--chunk", "65536"]):
        r = subprocess.run([exe, src, "--format", "fc32", "--out", out] + extra, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        raw, recs, o = open(out, "rb").read(), [], 0
        while o < len(raw):
            n = int.from_bytes(raw[o:o + 4], "little")
            recs.append(raw[o + 4:o + 4 + n])
            o += 4 + n
        assert recs == want, extra


def test_stream_longest_option_keeps_the_payload_list_and_cuts_what_is_longer(rx, po):
    """Option "stream_longest" (sizes the carry every batch re-synchronises): a stream whose frames all fit gives the reference chain's payload list
    for any batch size; a frame longer than the value that no buffer ever holds whole is missing from the list, and nothing else changes."""
    import fun_ofdm_amd as foa
    rng = np.random.default_rng(64)
    specs = [(int(rng.choice((5, 6, 8, 9, 10))), int(rng.integers(1, 700))) for _ in range(40)]       # <= 700 bytes at >= 18 Mbps: <= 320 + 80 * 80 samples
    iq, pays = _stream(po, rng, specs, snr_db=24.0, cfo_hz=3000.0, gap=(0, 400))
    want = po.ReceiverChain().run_stream(iq.astype(np.complex128))
    assert len(want) >= 36
    longest = 320 + 80 * 81 + 192
    rx.set_option("stream_longest", longest)
    try:
        for batch, chunk in ((4096, 4096), (8192, 1000), (65536, 7777), (1 << 18, 4096)):
            st = foa.Stream(rx, batch, narrow_threads=1)
            got = []
            try:
                for a in range(0, iq.size, chunk):
                    got += st.push(iq[a:a + chunk])
                got += st.flush()
            finally:
                st.close()
            assert got == want, (batch, chunk, len(got), len(want))
        # one frame longer than the value in the middle of the stream: it is the only one missing
        big = po.build_frame(rng.integers(0, 256, 3000, dtype=np.uint8), 5) * np.exp(1j * 0.7)
        sigma = np.sqrt(0.0124 / (2 * 10 ** 2.4))
        big = (big + (rng.normal(size=big.size) + 1j * rng.normal(size=big.size)) * sigma).astype(np.complex64)
        cut = iq.size // 2
        while np.abs(iq[cut - 50:cut + 50]).max() > 0.05:        # a quiet spot between two frames
            cut += 37
        iq2 = np.concatenate([iq[:cut], np.zeros(300, np.complex64), big, np.zeros(300, np.complex64), iq[cut:]])
        want2 = po.ReceiverChain().run_stream(iq2.astype(np.complex128))
        assert len(want2) == len(want) + 1
        st = foa.Stream(rx, 16384)
        try:
            got2 = st.push(iq2) + st.flush()
        finally:
            st.close()
        assert len(got2) == len(want) and [p for p in want2 if len(p) != 3000] == got2
        # ... with THESE batches: the value only sizes the carry, so whether a longer frame is delivered depends on whether a buffer ever
        # holds all of it (include/fun_ofdm_amd.h).  One 256 Ki-sample batch does: then the frame comes out like any other
        st = foa.Stream(rx, 1 << 18)
        try:
            got3 = st.push(iq2) + st.flush()
        finally:
            st.close()
        assert got3 == want2
        with pytest.raises(foa.FoaError):
            rx.set_option("stream_longest", 100)
    finally:
        rx.set_option("stream_longest", 0)
