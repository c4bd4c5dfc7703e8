"""Builds and runs tests/cpp/test_adaptors.cpp: the C++ block adaptors and receiver_chain::process_samples()
of include/fun_ofdm_amd/blocks.hpp against the oracle, through the C ABI.  GPU only."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
