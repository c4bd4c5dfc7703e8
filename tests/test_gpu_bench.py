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
    expect = 2 * 500 * 3520 / (out["ms_per_step"] * 1e-3) / 1e6
    assert abs(out["value"] - expect) <= 0.051 + expect * (0.5e-4 / out["ms_per_step"] + 1e-6)


def test_bench_over_every_device_count_of_this_box():
    """`bench.py --gpus N` for N = 1 .. the devices there are (one rank per device over RCCL, as the driver starts it on a node): N ranks in the
    collective, N x the frames decoded bit-exactly, RCCL reporting N ranks.  On a one-GPU box this is the N = 1 line; on an 8-GPU node it is
    BASELINE config 4's code path at reduced size, with nothing to edit."""
    import torch
    n_dev = torch.cuda.device_count()
    assert n_dev >= 1
    for n in range(1, n_dev + 1):
        e = dict(os.environ)
        e.pop("WORLD_SIZE", None)
        e["FOA_BENCH_FORCE_DIST"] = "1"                    # (N = 1 too goes through the process group and the gather)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--frames", "500", "--steps", "3", "--warmup", "2", "--no-extra-legs",
                            "--no-sync-leg", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=e)
        assert r.returncode == 0, r.stderr[-3000:]
        out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
        assert out["n_gpus"] == n and out["scaling"] == "weak" and out["config"]["psdu_bit_exact"] is True
        assert out["config"]["collective"]["ranks"] == n and out["config"]["collective"]["backend"] == "nccl" and out["config"]["collective"]["gathers_per_region"] == 3
        assert out["config"]["frames_ok"] >= n * 500 - n and ("RCCL saw %d rank" % n) in r.stderr


def test_bench_single_rank_line_has_the_contract_fields():
    out = _bench("--frames", "600", "--steps", "4", "--warmup", "2", "--legs-frames", "40", "--fill-frames", "96")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["config"]["psdu_bit_exact"] is True
    rf = out["roofline"]
    assert rf["bound"] == "valu" and 0 < rf["frac"] < 1 and rf["hbm"]["unit"] == "GB/s" and rf["kernel"] == "k_viterbi_fwd3"
    assert 70 < rf["peak"] < 85 and 0 < rf["frac"] <= rf["frac_at_step_rate"]["frac"] * 1.5      # frac: per launch (the tier's definition)
    assert abs(rf["frac"] - rf["algorithmic_ops_per_launch"] / (rf["avg_kernel_ms"] * 1e-3) / (rf["peak"] * 1e12)) < 2e-3
    live = rf["peak_measured_live"]                        # the issue probe ran in this process: packed instructions at ~4 clocks, plain VOP2 at ~2
    assert 3.5 < live["clk_per_packed_wave_instr"] < 4.8 and 1.8 < live["clk_per_plain_vop2_wave_instr"] < 2.6 and 1.5 < live["ghz"] < 2.6
    rp = out["repeats"]
    assert rp["regions"] == 3 and rp["min"] <= rp["median"] <= rp["max"] and rp["forward_live_over_alone"] > 0.5
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and len(cb["runs_s"]) == 3 and cb["cpu_model"]
    assert cb["timed_decoder_equals_scalar_checker"]["equal"] is True and cb["per_thread_single"]["value"] > 1.0 and 0 < cb["parallel_efficiency"] < 1.5
    assert cb["viterbi_ns_per_step"]["oracle_simd"] < cb["viterbi_ns_per_step"]["oracle_scalar_model"]
    legs = out["legs"]
    assert legs["end_to_end_host_pointers"]["same_results_as_device_path"] is True
    rows = legs["config3_rate_sweep"]["rates"]
    assert [r["rate_enum"] for r in rows] == [0, 2, 3, 5, 6, 8, 9, 10]
    assert all(r["psdu_bit_exact"] and r["crc_ok"] >= 30 for r in rows) and all(v for r in rows for k, v in r.items() if k.startswith("gpu_equals_cpu"))
    assert all(r["cpu_Msamples_per_s"] > 0 and r["cpu_crc_fail"] == r["gpu_crc_fail"] for r in rows)
    assert all(r["gpu_equals_cpu_on_all"] is True for r in rows) and legs["config3_rate_sweep"]["table"] == "small batch"
    fill = legs["config3_machine_filling"]["rates"]
    assert [r["rate_enum"] for r in fill] == [0, 2, 3, 5, 6, 8, 9, 10] and all(r["frames"] == 96 for r in fill)
    assert all(r["psdu_bit_exact"] and r["gpu_equals_cpu_on_all"] is True and r["scalar_checker_on_first"]["equal_to_gpu"] is True for r in fill)
    mixed = legs["config3_mixed_call"]
    assert mixed["frames"] == 8 * 40 and mixed["psdu_bit_exact"] is True and mixed["gpu_equals_cpu_on_all"] is True and mixed["crc_ok"] >= 8 * 40 - 12
    c5 = legs["config5_stream"]
    assert c5["psdu_bit_exact"] is True and c5["frames_ok"] >= 3900 and c5["gpu_equals_cpu_on_all"] is True
    assert c5["cpu_checked_alignments"] == c5["alignments"]
    p5 = c5["pipelined"]                                  # pre-sync in two halves (foa_rx_sync_dev_begin / _end) under the decode calls
    assert p5["same_descriptors_as_blocking_call"] is True and p5["same_results_as_blocking_call"] is True and p5["ms_sync_plus_decode"] > 0
    ws = out["config"]["incl_device_pre_sync"]
    assert ws["same_results_as_host_sync"] is True and ws["pipelined"]["same_results_as_host_sync"] is True
    ps = legs["process_samples_api"]
    assert ps["same_list_as_batch_path"] is True and ps["packets"] == ps["batch_path_payloads"]
    # no leg that is handed host buffers goes past the ceiling the line states for them (measured the way the engine copies)
    ceil = ps["pcie_ceiling"]
    assert ceil["h2d_GBps_measured"] > 5 and ps["Msamples_per_s"] / 1e3 <= ceil["Gsamples_per_s"] * 1.02 and ceil["leg_over_ceiling"] <= 1.02
    assert legs["end_to_end_host_pointers"]["Msamples_per_s"] / 1e3 <= ceil["Gsamples_per_s"] * 1.02
    # every rate of config 3 and the stream of config 5 carry their own roofline: the bounding kernel, its roof and the fraction reached
    for r in rows + fill + [mixed, c5]:
        q = r["roofline"]
        assert q["bound"] in ("valu", "hbm") and q["kernel"] in q["kernels"] and 0 < q["frac"] < 1 and q["avg_kernel_ms"] > 0 and q["trellis_steps"] > 0
        k = q["kernels"][q["kernel"]]
        assert abs(q["frac"] - k["algorithmic"] / (k["ms_alone"] * 1e-3) / ({"GB/s": 1e9, "T ops/s": 1e12}[k["unit"]] * k["peak"])) < 5e-3
        assert set(q["kernels"]) >= {"k_data_symbols_q4", "k_viterbi_fwd3", "k_tb_walk + k_tb_finish"}
    assert "k_sync_*" in c5["roofline"]["kernels"] and all(r["max_dbps"] == {0: 24, 2: 36, 3: 48, 5: 72, 6: 96, 8: 144, 9: 192, 10: 216}[r["rate_enum"]] for r in fill)
    assert rf["frac_step_rate"] == rf["frac_at_step_rate"]["frac"] and rf.get("launch_ms", 1.0) > 0      # (launch_*: only with enough steps to read spacings)
    if "valu_busy_call_frac_of_step" in rf:                        # (counts from profiles/ at this workload size; the clock and the step live)
        assert 0.3 < rf["valu_busy_call_frac_of_step"] < 1.1 and rf["valu_busy_call_ms"] == rf["valu_call"]["busy_ms"]


def test_bench_forced_collective_path_over_rccl_with_one_rank():
    """FOA_BENCH_FORCE_DIST=1: the N > 1 host path -- `nccl` process group, three rotating output sets, wait_age(2), read_done
    events, one dist.gather of device uint8 tensors per step (fun_ofdm_amd/shard.py) -- with the one rank a one-GPU box has.
    RCCL itself runs; config 4 (8 GPUs) stays unmeasured until a node exists."""
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e["FOA_BENCH_FORCE_DIST"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "500", "--steps", "4", "--warmup", "3", "--no-extra-legs", "--no-sync-leg",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=e)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["config"]["psdu_bit_exact"] is True
    assert "RCCL" in out["config"]["sharding"] and out["config"]["collective"] == {"backend": "nccl", "ranks": 1, "gathers_per_region": 4}
    assert "RCCL saw 1 rank" in r.stderr


def test_rccl_gather_of_device_psdus_world_one():
    """The very call shard.gather_psdus makes at N > 1 -- dist.gather of a device uint8 tensor -- over RCCL with one rank."""
    import torch
    import torch.distributed as dist
    from fun_ofdm_amd import shard
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29713")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        x = (torch.arange(7 * 1024, device=dev) % 251).to(torch.uint8).reshape(7, 1024)
        g = shard.gather_psdus(x, 7, 0, 1, force_collective=True)
        torch.cuda.synchronize()
        assert g.is_cuda and g.shape == (7, 1024) and bool(torch.equal(g, x))
    finally:
        dist.destroy_process_group()
