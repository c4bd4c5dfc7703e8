"""The drop-in boundary against the reference's OWN types and blocks, in the dev container (CPU; skipped where /root/reference is absent):
include/fun_ofdm_amd/blocks.hpp compiled AFTER the reference's block.h / tagged_vector.h at -std=c++11 (the integration mode INTEGRATION.md
documents), real fun::frame_detector / fun::timing_sync objects in front of the repository's blocks, wired as fun::receiver_chain wires
them (tests/cpp/mixed_chain.cpp).  No GPU here: the C ABI is answered by tests/cpp/stub_abi.cpp through the oracle, so what runs is the
adaptors' host logic under the reference's types; the same program runs against the real library in tests/test_gpu_cpp_adaptors.py."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def test_blocks_hpp_under_the_reference_headers_with_real_presync_blocks(tmp_path, po):
    ref_lib = os.path.join(ROOT, "oracle", "_ref", "libfun_ofdm_ref.so")
    if not (os.path.exists(os.path.join(REF, "src", "block.h")) and os.path.exists(ref_lib)):
        pytest.skip("needs the reference tree (/root/reference) and its partial build oracle/_ref: dev container only")
    ora = os.path.join(ROOT, "oracle")
    stub_o, exe = str(tmp_path / "stub_abi.o"), str(tmp_path / "mixed_chain")
    subprocess.run(["g++", "-O1", "-std=c++17", "-c", os.path.join(ROOT, "tests", "cpp", "stub_abi.cpp"), "-I", os.path.join(ROOT, "include"), "-I", ora,
                    "-o", stub_o], check=True)
    # the reference's language level and its headers FIRST on the include path of a translation unit that includes them first
    subprocess.run(["g++", "-O1", "-std=c++11", "-DFOA_REFERENCE_HEADERS", "-I", os.path.join(REF, "src"), os.path.join(ROOT, "tests", "cpp", "mixed_chain.cpp"),
                    stub_o, "-I", os.path.join(ROOT, "include"), "-I", ora, "-L", ora, "-loracle", "-L", os.path.dirname(ref_lib), "-lfun_ofdm_ref",
                    "-Wl,-rpath," + ora, "-Wl,-rpath," + os.path.dirname(ref_lib), "-lm", "-lpthread", "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "built against the reference's own block.h" in r.stdout, r.stdout + r.stderr
