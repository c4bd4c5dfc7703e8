"""The N>1 path's host logic on CPU: round-robin frame shards and the PSDU gather, world_size 2 over gloo."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_psdu(global_ids, slot):
    g = global_ids.to(torch.int64)
    col = torch.arange(slot, dtype=torch.int64)
    return ((g[:, None] * 131 + col[None, :] * 7 + 3) % 251).to(torch.uint8)


def _worker(rank, world, port, n_global, slot, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fun_ofdm_amd import shard
    ids = shard.local_frame_ids(n_global, rank, world)
    out = shard.gather_psdus(_fake_psdu(ids, slot), n_global, rank, world)
    # the same through receive buffers allocated once, twice over (reuse), read in global order through the strided view
    bufs = shard.GatherBuffers(n_global, slot, world, torch.device("cpu")) if rank == 0 else None
    views = []
    for rep in range(2):
        v = shard.gather_psdus((_fake_psdu(ids, slot) + rep).to(torch.uint8), n_global, rank, world, buffers=bufs, materialize=False)
        views.append(None if v is None else v.reshape(-1, slot)[:n_global].clone())
    mat = shard.gather_psdus(_fake_psdu(ids, slot), n_global, rank, world, buffers=bufs)
    dist.barrier()
    if rank == 0:
        want = _fake_psdu(torch.arange(n_global), slot)
        ok = bool(torch.equal(out, want)) and out.shape == (n_global, slot)
        ok = ok and bool(torch.equal(views[0], want)) and bool(torch.equal(views[1], (want + 1).to(torch.uint8))) and bool(torch.equal(mat, want))
        q.put(ok)
    else:
        assert out is None and views == [None, None] and mat is None
    dist.destroy_process_group()


def _run(world, n_global, slot):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_global, slot, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return ok


def _worker_slots(rank, world, port, frames, slot, q):
    """The gather bench.py makes since round 6: every rank's output set as the decode wrote it -- one row per ALIGNMENT, more rows than frames,
    a rank-specific frame -> row map -- gathered as it is; rank 0 forms the global frame order from the maps it was handed once."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fun_ofdm_amd import shard
    n_global = frames * world
    ids = shard.local_frame_ids(n_global, rank, world)
    g = torch.Generator().manual_seed(100 + rank)
    rows = frames + 3 + rank                                 # this rank's alignments (noise alignments among them), + the zero row
    t_rows = torch.tensor([rows + 1])
    dist.all_reduce(t_rows, op=dist.ReduceOp.MAX)
    R = int(t_rows.item())
    perm = torch.randperm(rows, generator=g)[:frames]         # local frame k sits in row perm[k]
    miss = frames // 2 if rank == 1 else -1                   # a frame the detector missed -> the zero row
    if miss >= 0:
        perm[miss] = rows
    out = torch.full((R, slot), 0xEE, dtype=torch.uint8)
    out[rows:] = 0
    want_local = _fake_psdu(ids, slot)
    for k in range(frames):
        if k != miss:
            out[perm[k]] = want_local[k]
    perms = shard.gather_maps(perm, rank, world)
    bufs = shard.SlotBuffers(R, slot, world, torch.device("cpu")) if rank == 0 else None
    ok = True
    for rep in range(2):                                      # receive buffers are reused from step to step
        parts = shard.gather_slots((out + rep).to(torch.uint8) if rep else out, rank, world, buffers=bufs)
        if rank == 0:
            got = shard.order_gathered(parts, perms, n_global)
            want = _fake_psdu(torch.arange(n_global), slot)
            if world > 1:
                want[1 + (frames // 2) * world] = 0            # rank 1's missed frame
            if rep:
                want = (want + 1).to(torch.uint8)
            ok = ok and got.shape == (n_global, slot) and bool(torch.equal(got, want))
        else:
            assert parts is None and perms is None
    dist.barrier()
    if rank == 0:
        q.put(ok)
    dist.destroy_process_group()


def test_gather_of_output_sets_as_decoded_rank0_orders_afterwards():
    for world, frames in ((2, 9), (3, 5)):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker_slots, args=(r, world, port, frames, 16, q)) for r in range(world)]
        for p in procs:
            p.start()
        assert q.get(timeout=120)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0


def test_round_robin_ids_partition_the_frames():
    from fun_ofdm_amd import shard
    for world in (1, 2, 3, 8):
        for n in (0, 1, 7, 64, 1001):
            ids = torch.cat([shard.local_frame_ids(n, r, world) for r in range(world)])
            assert sorted(ids.tolist()) == list(range(n))


def test_gather_two_ranks_even_and_ragged():
    assert _run(2, 10, 16)          # even split
    assert _run(2, 11, 32)          # ragged: rank 0 has one frame more


def test_single_rank_is_identity():
    from fun_ofdm_amd import shard
    x = _fake_psdu(torch.arange(5), 8)
    assert shard.gather_psdus(x, 5, 0, 1) is x
