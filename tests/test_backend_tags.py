"""fun_amd::rx_backend (include/fun_ofdm_amd/blocks.hpp) over PLACED tags, cut into work() calls of several sizes: the payloads it hands on
against the oracle's block chain over the same tag stream (fft_symbols.cpp:41-73, channel_est.cpp:44-85, frame_decoder.cpp:52-88).  The
cases are those of tests/manual/stress_tags.py: second preambles late in a frame, pile-ups of LTS1 tags less than 64 samples apart (the later
alignment's vectors move one symbol on), cut LTS windows.  A pile-up that straddles two work() calls must come out as in one call: the block
keeps the alignments of a pile-up together until the last of them is decided.
No GPU here: the C ABI is answered by tests/cpp/stub_abi.cpp through the oracle, so what runs is the block's host logic; the same program
runs against the real library in tests/test_gpu_cpp_adaptors.py."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "manual"))
LTS1, LTS2 = 4, 5           # fun::vector_tag (src/tagged_vector.h:24-33)


def tag_bytes(n, descs):
    """timing_sync.cpp:105-106 for every alignment in stream order: a later tag on the same sample replaces the earlier one"""
    t = np.zeros(n, np.uint8)
    for p in descs["lts1_pos"]:
        t[int(p)] = LTS1
        if int(p) + 64 < n:
            t[int(p) + 64] = LTS2
    return t


def write_case(tmp_path, seed):
    import stress_tags as st
    s, d = st.make_case(seed)
    f_s, f_t = str(tmp_path / ("s%d.f64" % seed)), str(tmp_path / ("t%d.u8" % seed))
    st.rotated(s, d).astype(np.complex128).tofile(f_s)
    tag_bytes(s.size, d).tofile(f_t)
    return s, d, f_s, f_t


def build_stub_program(tmp_path):
    ora = os.path.join(ROOT, "oracle")
    exe = str(tmp_path / "backend_tags")
    subprocess.run(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "backend_tags.cpp"), os.path.join(ROOT, "tests", "cpp", "stub_abi.cpp"),
                    "-I", os.path.join(ROOT, "include"), "-I", ora, "-L", ora, "-loracle", "-Wl,-rpath," + ora, "-lm", "-lpthread", "-o", exe], check=True)
    return exe


def run_program(exe, f_s, f_t, chunk):
    r = subprocess.run([exe, f_s, f_t, str(chunk)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return [bytes.fromhex(line) for line in r.stdout.split()]


def test_rx_backend_over_placed_tags_in_calls_of_any_size(tmp_path, po):
    exe = build_stub_program(tmp_path)
    piles = 0
    for seed in range(24):
        s, d, f_s, f_t = write_case(tmp_path, seed)
        piles += int(np.sum(np.diff(d["lts1_pos"]) < 64))
        want = po.chain_from_tags_f32(s, d)
        for chunk in (4096, 997, 61):
            got = run_program(exe, f_s, f_t, chunk)
            assert got == want, (seed, chunk, len(got), len(want))
    assert piles >= 5
