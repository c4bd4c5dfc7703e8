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
