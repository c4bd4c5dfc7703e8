"""bench.py as the driver runs it, at reduced size, on the GPU box: the single-rank line and the N = 2 path (two rank
processes started by bench.py itself; on a one-GPU box they share the device and the PSDU gather goes over gloo)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True, timeout=900, env=e)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]              # ONE JSON line
    return json.loads(lines[0])


def test_bench_two_ranks_decode_and_gather():
    out = _bench("--gpus", "2", "--frames", "500", "--steps", "3", "--warmup", "2", "--no-extra-legs", "--no-sync-leg")
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 3
    cfg = out["config"]
    assert cfg["psdu_bit_exact"] is True and cfg["gpu_equals_cpu_on_sample"] is True
    assert cfg["frames_ok"] >= 2 * 500 - 2 and "rank i mod 2" in cfg["sharding"]
    assert abs(out["value"] - 2 * 500 * 3520 / (out["ms_per_step"] * 1e-3) / 1e6) / out["value"] < 1e-3


def test_bench_single_rank_line_has_the_contract_fields():
    out = _bench("--frames", "600", "--steps", "4", "--warmup", "2", "--legs-frames", "40")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["config"]["psdu_bit_exact"] is True
    rf = out["roofline"]
    assert rf["bound"] == "valu" and 0 < rf["frac"] < 1 and rf["hbm"]["unit"] == "GB/s" and rf["kernel"] == "k_viterbi_fwd3"
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    legs = out["legs"]
    assert legs["end_to_end_host_pointers"]["same_results_as_device_path"] is True
    rows = legs["config3_rate_sweep"]["rates"]
    assert [r["rate_enum"] for r in rows] == [0, 2, 3, 5, 6, 8, 9, 10]
    assert all(r["psdu_bit_exact"] and r["crc_ok"] >= 30 for r in rows) and all(v for r in rows for k, v in r.items() if k.startswith("gpu_equals_cpu"))
    c5 = legs["config5_stream"]
    assert c5["psdu_bit_exact"] is True and c5["frames_ok"] >= 3900
