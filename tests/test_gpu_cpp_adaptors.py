"""Builds and runs tests/cpp/test_adaptors.cpp: the C++ block adaptors and receiver_chain::process_samples()
of include/fun_ofdm_amd/blocks.hpp against the oracle, through the C ABI.  GPU only."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def test_cpp_adaptors(tmp_path, po):
    import fun_ofdm_amd as foa
    exe = str(tmp_path / "test_adaptors")
    libdir = os.path.dirname(foa.library_path())
    cmd = ["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "test_adaptors.cpp"), "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "oracle"), "-L", libdir, "-lfun_ofdm_amd", "-L", os.path.join(ROOT, "oracle"), "-loracle",
           "-Wl,-rpath," + libdir, "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-Wl,-rpath,/opt/rocm/lib", "-lm", "-lpthread", "-o", exe]
    subprocess.run(cmd, check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr


def test_device_side_ordering_against_the_callers_own_streams(tmp_path, po):
    """tests/cpp/test_ordering.cpp: a capture loop with a device-side producer on the caller's own normal-priority streams and NO host
    synchronisation between producing the samples and decoding them (foa_rx_decode_frames_dev_after, foa_rx_sync_dev_begin_after,
    foa_rx_record_consumed, foa_rx_record_done) -- 200 pipelined rounds, 60 in line, 60 with the pre-sync in the loop, each round equal to
    the oracle's decode of that round's capture."""
    import fun_ofdm_amd as foa
    exe = str(tmp_path / "test_ordering")
    libdir, ora = os.path.dirname(foa.library_path()), os.path.join(ROOT, "oracle")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "test_ordering.cpp"), "-I", os.path.join(ROOT, "include"), "-I", ora,
                    "-L", libdir, "-lfun_ofdm_amd", "-L", ora, "-loracle", "-Wl,-rpath," + libdir, "-Wl,-rpath," + ora, "-lm", "-lpthread", "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


def test_mixed_chain_reference_presync_blocks_in_front_of_the_gpu_blocks(tmp_path, po):
    """tests/cpp/mixed_chain.cpp against the real library: the reference's compiled frame_detector and timing_sync (oracle/_ref, built
    from /root/reference in the dev container; the library travels) -> fun_amd::fft_symbols .. frame_decoder, and -> the fused
    fun_amd::rx_backend, wired as fun::receiver_chain wires its blocks (src/receiver_chain.cpp:29-51, :106-126): the oracle chain's
    ordered payload list on a mixed-rate stream with CFO.  (The same file is compiled against the reference's own headers in
    tests/test_boundary_reference.py.)"""
    import fun_ofdm_amd as foa
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    if not os.path.exists(os.path.join(ref_dir, "libfun_ofdm_ref.so")):
        pytest.skip("oracle/_ref/libfun_ofdm_ref.so is not on this box (built from /root/reference in the dev container, git-ignored)")
    exe = str(tmp_path / "mixed_chain")
    libdir, ora = os.path.dirname(foa.library_path()), os.path.join(ROOT, "oracle")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "mixed_chain.cpp"), "-I", os.path.join(ROOT, "include"), "-I", ora,
                    "-L", libdir, "-lfun_ofdm_amd", "-L", ora, "-loracle", "-L", ref_dir, "-lfun_ofdm_ref", "-Wl,-rpath," + libdir, "-Wl,-rpath," + ora,
                    "-Wl,-rpath," + ref_dir, "-Wl,-rpath,/opt/rocm/lib", "-lm", "-lpthread", "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "chain B" in r.stdout, r.stdout + r.stderr


def test_rx_backend_over_placed_tags_in_calls_of_any_size(tmp_path, po):
    """tests/cpp/backend_tags.cpp against the real library: fun_amd::rx_backend over the placed-tag cases of tests/manual/stress_tags.py
    (second preambles inside frames, pile-ups of LTS1 tags down to one sample apart), cut into work() calls of several sizes, against the
    oracle's block chain over the same tags (the CPU twin over the stub ABI: tests/test_backend_tags.py)."""
    import fun_ofdm_amd as foa
    import test_backend_tags as t
    exe = str(tmp_path / "backend_tags")
    libdir = os.path.dirname(foa.library_path())
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "backend_tags.cpp"), "-I", os.path.join(ROOT, "include"),
                    "-L", libdir, "-lfun_ofdm_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lm", "-lpthread", "-o", exe], check=True)
    piles = n = 0
    for seed in range(1000, 1040):
        s, d, f_s, f_t = t.write_case(tmp_path, seed)
        piles += int(np.sum(np.diff(d["lts1_pos"]) < 64))
        want = po.chain_from_tags_f32(s, d)
        n += len(want)
        for chunk in (4096, 997):
            assert t.run_program(exe, f_s, f_t, chunk) == want, (seed, chunk)
    assert piles >= 8 and n >= 15, (piles, n)


def test_sim_cli_over_iq_file(tmp_path):
    """examples/foa_sim.cpp: a raw fc32 capture in, length-prefixed PSDU records out, through
    fun_amd::file_source -> fun_amd::receiver -> receiver_chain::process_samples."""
    import numpy as np
    import fun_ofdm_amd as foa
    from fun_ofdm_amd import synth
    exe = str(tmp_path / "foa_sim")
    libdir = os.path.dirname(foa.library_path())
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "examples", "foa_sim.cpp"), "-I", os.path.join(ROOT, "include"),
                    "-L", libdir, "-lfun_ofdm_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", exe], check=True)
    pays = synth.splitmix64_bytes(0xC11, 24, 300)
    iq, _ = synth.make_stream(synth.build_frames(pays, 8), 4096, 200, 25.0, seed=9)
    for fmt, data, extra in (("fc32", iq.astype(np.complex64), []), ("fc64", iq.astype(np.complex128), []),
                             ("fc32", iq.astype(np.complex64), ["--async", "4"])):
        src, out = str(tmp_path / ("cap." + fmt)), str(tmp_path / ("psdus." + fmt))
        data.tofile(src)
        r = subprocess.run([exe, src, "--format", fmt, "--out", out, "--chunk", "4096"] + extra, capture_output=True, text=True, timeout=600)
        print(r.stdout, r.stderr)
        assert r.returncode == 0, r.stdout + r.stderr
        raw = open(out, "rb").read()
        got, o = [], 0
        while o < len(raw):
            n = int.from_bytes(raw[o:o + 4], "little")
            got.append(raw[o + 4:o + 4 + n])
            o += 4 + n
        assert o == len(raw)
        sent = [p.tobytes() for p in pays]
        assert len(got) >= 22 and all(g in sent for g in got)
        assert [sent.index(g) for g in got] == sorted(sent.index(g) for g in got)      # stream order
        assert ("%d packets" % len(got)) in r.stdout


def _collision_stream(po, case, seed, ga=0.15, snr=40.0):
    """A weak 6 Mbps frame A (300 bytes, 8640 samples) with a second, nominal-level frame B (12 Mbps, 100 bytes) starting in the
    middle of A's data symbols -- B's SIGNAL intact ("valid"), garbled ("invalid") or B absent ("none") -- then a clean frame C."""
    import numpy as np
    rng = np.random.default_rng(seed)
    pay = [rng.integers(0, 256, n, dtype=np.uint8) for n in (300, 100, 100)]
    A, B, C = po.build_frame(pay[0], 0) * ga, po.build_frame(pay[1], 3), po.build_frame(pay[2], 8)
    if case == "invalid":
        B = B.copy()
        B[320:400] = B[320:400][::-1] * 1j
    s = np.zeros(400 + A.size + 600 + C.size + 500, complex)
    s[400:400 + A.size] += A
    off = 400 + 320 + 80 * 40 + 37
    if case != "none":
        s[off:off + B.size] += B
    s[400 + A.size + 600:400 + A.size + 600 + C.size] += C
    s = s + (rng.normal(size=s.size) + 1j * rng.normal(size=s.size)) * np.sqrt(0.0124 / 2 / 10 ** (snr / 10))
    return s.astype(np.complex64), [p.tobytes() for p in pay]


def test_second_preamble_inside_a_frame(tmp_path, po):
    """frame_decoder.cpp:52-76: a preamble that arrives before the current frame ends.  With a valid SIGNAL the reference
    abandons the frame it was collecting and starts the new one; with an invalid SIGNAL it keeps filling the old frame (from
    the re-aligned symbol windows, fft_symbols.cpp:41-50), which then fails its CRC.  Either way the old frame is never
    delivered.  The batch path does the same with the linked alignments (FOA_ST_SUPERSEDED / FOA_ST_CRC_FAIL, result for result the
    oracle's restatement of those rules) and must deliver exactly the reference chain's ordered payload list; so must
    fun_amd::receiver_chain::process_samples in both of its modes."""
    import numpy as np
    import fun_ofdm_amd as foa
    exe = str(tmp_path / "foa_sim")
    libdir = os.path.dirname(foa.library_path())
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "examples", "foa_sim.cpp"), "-I", os.path.join(ROOT, "include"),
                    "-L", libdir, "-lfun_ofdm_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", exe], check=True)
    rx = foa.Receiver(0)
    seen = set()
    try:
        for seed in (5, 7):
            for case in ("valid", "invalid", "none"):
                iq, pays = _collision_stream(po, case, seed)
                want = po.ReceiverChain().run_stream(iq.astype(np.complex128))          # the reference-shaped chain (oracle)
                names = [{pays[0]: "A", pays[1]: "B", pays[2]: "C"}.get(x, "?") for x in want]
                assert names == {"valid": ["B", "C"], "invalid": ["C"], "none": ["A", "C"]}[case], (seed, case, names)
                # batch path: per-alignment results as the oracle's alignment decoder has them, payload list as the chain's
                descs = foa.find_alignments(iq)
                assert descs.tobytes() == po.find_alignments_f32(iq).tobytes()
                ends = foa.alignment_ends(descs, iq.size)
                psdu, res = rx.decode_frames_host(iq, descs, ends)
                opsdu, ores = po.decode_batch_f32(iq, descs, ends)
                assert np.array_equal(res.view(np.int32), ores.view(np.int32)), (seed, case)
                got = [psdu[f, :res[f]["length"]].tobytes() for f in range(descs.size) if res[f]["status"] == foa.ST_OK]
                assert got == want, (seed, case)
                if case != "none":
                    # frame A: abandoned when B's valid SIGNAL arrives; with B's SIGNAL garbled it fills on with the re-aligned symbols and fails its CRC
                    assert descs.size == 3 and res[0]["status"] == (foa.ST_SUPERSEDED if case == "valid" else foa.ST_CRC_FAIL)
                    assert res[0]["rate"] == 0 and res[0]["length"] == 300
                    assert res[1]["status"] == (foa.ST_OK if case == "valid" else foa.ST_HEADER_FAIL)
                seen.add((case, tuple(int(x) for x in res["status"])))
                # fun_amd::receiver_chain (synchronous and in asynchronous batches) over the same capture
                src, out = str(tmp_path / "cap.fc32"), str(tmp_path / "psdus")
                iq.tofile(src)
                for extra in ([], ["--async", "2"]):
                    r = subprocess.run([exe, src, "--format", "fc32", "--out", out, "--chunk", "4096"] + extra, capture_output=True, text=True, timeout=300)
                    assert r.returncode == 0, r.stdout + r.stderr
                    raw, recs, o = open(out, "rb").read(), [], 0
                    while o < len(raw):
                        n = int.from_bytes(raw[o:o + 4], "little")
                        recs.append(raw[o + 4:o + 4 + n])
                        o += 4 + n
                    assert recs == want, (seed, case, extra)
    finally:
        rx.close()
    assert len(seen) >= 3
