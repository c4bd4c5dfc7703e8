"""ASan + UBSan and TSan over the oracle (threaded chain, threaded batch decoder) and the host side of the drop-in
(blocks.hpp receiver_chain / receiver / sources, sync_host.h) linked against a stub C ABI -- tools/run_sanitizers.sh.
CPU only (the GPU pool offers no sanitizers); the recorded run of the round is profiles/*_sanitizers.txt."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc with libasan / libtsan")
def test_host_code_and_oracle_are_clean_under_sanitizers(tmp_path):
    out = str(tmp_path / "sanitizers.txt")
    r = subprocess.run([os.path.join(ROOT, "tools", "run_sanitizers.sh"), out], capture_output=True, text=True, timeout=900)
    text = open(out).read()
    assert r.returncode == 0 and "overall: clean" in text, text[-4000:]
    assert text.count("OK") >= 2 and "sanitizer reports: 0" in text
