"""bench.py's N > 1 host logic on CPU: two ranks over gloo run the REAL bench.run() -- workload generation, host pre-sync,
rotating output sets, wait_age / gather of the step two back, the final gathers, the checks and the JSON record -- with a
stand-in for the receiver handle (no GPU here).  The stand-in "decodes" by copying the payload each alignment's frame was
built from, and deliberately completes a step's outputs only when the library would (at the next call, at wait_age or
at sync), so that a gather that reads an output set too early, or a step that reuses one too soon, shows up as a mismatch.
Also: the --gpus launcher starts its ranks itself and refuses a --gpus / WORLD_SIZE mismatch."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class StubReceiver:
    """What bench.run() needs of fun_ofdm_amd.Receiver.  decode_frames_dev() only QUEUES a call; its outputs are written
    when the real library guarantees them: at wait_age(age) for the calls `age` back and older, at sync(), or when the call
    six back has to be complete because its work set is taken again (six rotating sets)."""
    pays = None        # set per rank: uint8[n_frames, 1024], payload of local frame i
    pitch, first = 4096, 176 + 184

    def __init__(self, device=0):
        self.queue = []
        self.calls = 0

    def set_option(self, name, value):
        pass

    def reserve(self, n_samples, n_frames):
        pass

    def _complete(self, job):
        descs, psdu, res = job
        d = descs.numpy().view(np.dtype([("lts1_pos", np.int64), ("rot_start", np.int64), ("c", np.float64), ("s", np.float64),
                                         ("c_prev", np.float64), ("s_prev", np.float64)]))
        on = (d["lts1_pos"] - self.first) % self.pitch == 0
        which = (d["lts1_pos"] - self.first) // self.pitch
        r = np.zeros((d.size, 4), np.int32)
        r[:, 0] = 1
        r[:, 1] = -1
        r[on] = (0, 10, 1024, 39)
        p = np.zeros((d.size, 1024), np.uint8)
        p[on] = self.pays[which[on]]
        psdu.copy_(torch.from_numpy(p))
        res.copy_(torch.from_numpy(r))

    def decode_frames_dev(self, iq, descs, ends, psdu, results, n_context=0, n_lead=0, settle=True):
        psdu.fill_(0xEE)                                   # in flight: whoever reads this set now reads garbage
        self.queue.append((descs, psdu, results))
        self.calls += 1
        while len(self.queue) > 5:
            self._complete(self.queue.pop(0))

    def wait_age(self, age):
        while len(self.queue) > age:
            self._complete(self.queue.pop(0))

    def sync(self):
        self.wait_age(0)

    def kernel_ms(self, previous=False, age=None):
        return dict(header=0.0, scan=0.0, symbols=0.0, viterbi_fwd=1.0, viterbi_finish=0.0, total=1.0)

    def close(self):
        pass


def _worker(rank, world, port, frames, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import bench
    from fun_ofdm_amd import shard, synth
    args = bench.parse_args(["--gpus", str(world), "--frames", str(frames), "--steps", str(steps), "--warmup", "2", "--no-cpu-baseline", "--no-extra-legs",
                             "--no-sync-leg"])
    ids = shard.local_frame_ids(frames * world, rank, world).numpy()
    StubReceiver.pays = synth.splitmix64_bytes(bench.SEED_BASE, len(ids), bench.PAYLOAD, ids=ids)
    out = bench.run(args, rank, world, rank, backend="gloo", make_receiver=StubReceiver, on_cpu=True)
    import torch.distributed as dist
    dist.barrier()
    if rank == 0:
        q.put(out)
    else:
        assert out is None
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 8], ids=["two-ranks", "eight-ranks"])
def test_bench_host_logic_two_ranks_over_gloo(world):
    """(world 8 = the rank count of BASELINE config 4; the name is kept from the two-rank version)"""
    frames, steps = 24, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, frames, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    json.dumps(out)                                        # the record is one JSON-serialisable object
    assert out["n_gpus"] == world and out["steps"] == steps and out["scaling"] == "weak" and out["vs_baseline"] is None
    cfg = out["config"]
    assert cfg["psdu_bit_exact"] is True                   # incl. the gathered slots of both ranks, in global frame order
    assert cfg["frames_ok"] == world * frames and cfg["frames_per_gpu"] == frames
    assert ("rank i mod %d" % world) in cfg["sharding"] and "gloo" in cfg["sharding"]
    # value counts the frames of ALL ranks (whole-job throughput)
    # (`value` is rounded to 0.1 Msample/s and `ms_per_step` to 1e-4 ms in the record: allow exactly that much)
    expect = world * frames * 3520 / (out["ms_per_step"] * 1e-3) / 1e6
    assert abs(out["value"] - expect) <= 0.051 + expect * (0.5e-4 / out["ms_per_step"] + 1e-6)
    rp = out["repeats"]
    assert rp["regions"] == 3 and len(rp["ms_per_step"]) == 3 and rp["min"] <= rp["median"] <= rp["max"] and rp["median"] == out["ms_per_step"]


def test_gpus_flag_must_match_world_size():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


def test_launcher_starts_the_ranks_itself(tmp_path):
    """`python bench.py --gpus 2` with WORLD_SIZE unset: two child ranks with RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, started
    before this process imports torch; their exit status is the launcher's.  (The children here are a recording stand-in
    for bench.py itself: there is no GPU to run the real ranks on.)"""
    sys.path.insert(0, ROOT)
    import bench
    probe = tmp_path / "probe.py"
    probe.write_text("import os, sys\n"
                     "open(os.path.join(%r, 'rank' + os.environ['RANK']), 'w').write(' '.join([os.environ['WORLD_SIZE'], os.environ['LOCAL_RANK'],"
                     " os.environ['MASTER_ADDR'], os.environ['MASTER_PORT']] + sys.argv[1:]))\n"
                     "sys.exit(3 if os.environ['RANK'] == '1' and '--fail' in sys.argv else 0)\n" % str(tmp_path))
    real = bench.__file__
    env_before = os.environ.pop("WORLD_SIZE", None)
    try:
        bench.__file__ = str(probe)
        assert bench.launch_ranks(2, ["--gpus", "2", "--steps", "1"]) == 0
        got = [(tmp_path / ("rank%d" % r)).read_text().split() for r in range(2)]
        assert got[0][0] == got[1][0] == "2" and [g[1] for g in got] == ["0", "1"] and got[0][2] == "127.0.0.1" and got[0][3] == got[1][3]
        assert got[0][4:] == ["--gpus", "2", "--steps", "1"]
        assert bench.launch_ranks(2, ["--fail"]) == 3      # a failing rank fails the launch
    finally:
        bench.__file__ = real
        if env_before is not None:
            os.environ["WORLD_SIZE"] = env_before
