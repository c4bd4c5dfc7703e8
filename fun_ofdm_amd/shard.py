"""Frame sharding across GPUs (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI).

Frames are independent, so the data path has no collective: global frame i lives on rank i mod G
(BASELINE config 4) and is decoded there; the only exchange is one gather of the decoded PSDU slots to rank 0.
"""
import torch
import torch.distributed as dist


def local_frame_ids(n_global, rank, world):
    """Global indices of the frames rank `rank` decodes (round-robin)."""
    return torch.arange(rank, max(n_global, rank), world)


class GatherBuffers:
    """Receive buffers of gather_psdus on rank 0, allocated once and reused from step to step.  On a GPU (RCCL) they are the
    `world` slabs uint8[m_max, slot] of ONE uint8[world, m_max, slot] tensor, so that the gathered slots can be read in GLOBAL
    frame order through a strided view (frame r + k * world = big[r, k]) without another copy.  In host memory (the gloo
    stand-in of the CPU tests) they are separate tensors -- gloo's gather into slabs of one tensor was measured 20 x slower than
    into tensors of their own (95 against 4 ms for 2 x 10 MB) -- and the global order is formed by a copy."""

    def __init__(self, n_global, slot, world, device, dtype=torch.uint8):
        self.m_max = (n_global + world - 1) // world
        device = torch.device(device)
        if device.type == "cpu":
            self.big = None
            self.parts = [torch.empty((self.m_max, slot), dtype=dtype) for _ in range(world)]
        else:
            self.big = torch.empty((world, self.m_max, slot), dtype=dtype, device=device)
            self.parts = list(self.big.unbind(0))

    def view_global(self):
        """[m_max, world, slot]: index [k, r] = global frame r + k * world (a view of the receive buffer on a GPU)."""
        return self.big.transpose(0, 1) if self.big is not None else torch.stack(self.parts, dim=1)


def gather_psdus(psdu_local, n_global, rank, world, group=None, force_collective=False, buffers=None, materialize=True):
    """psdu_local: uint8[m_local, slot] for this rank's frames in local order (global ids rank, rank+G, ...).
    Returns on rank 0 a uint8[n_global, slot] tensor in GLOBAL frame order, None elsewhere.  One gather.
    force_collective: make the dist.gather call at world 1 too (a one-GPU box then exercises the very RCCL call the
    8-GPU job makes; needs an initialised process group).
    buffers: a GatherBuffers to receive into on rank 0 (no allocation per call); materialize=False then returns the strided
    [m_max, world, slot] view of it -- global frame order without the reordering copy (80 MB per step on rank 0 at 8 x 10 000
    frames), to be flattened by whoever needs the contiguous tensor."""
    if world == 1 and not force_collective:
        return psdu_local
    slot = psdu_local.shape[1]
    m_max = (n_global + world - 1) // world
    pad = psdu_local
    if psdu_local.shape[0] != m_max:
        pad = torch.zeros((m_max, slot), dtype=psdu_local.dtype, device=psdu_local.device)
        pad[:psdu_local.shape[0]] = psdu_local
    if rank == 0:
        parts = buffers.parts if buffers is not None else [torch.empty_like(pad) for _ in range(world)]
    else:
        parts = None
    dist.gather(pad, parts, dst=0, group=group)
    if rank != 0:
        return None
    if buffers is not None and not materialize:
        return buffers.view_global()
    # parts[r][k] is global frame r + k*world: interleave
    stacked = torch.stack(parts, dim=1).reshape(m_max * world, slot)
    return stacked[:n_global]


# ---- the same gather without a reordering pass in the timed loop -------------------------------------------------------------------------
# A rank's decode call writes one PSDU slot per ALIGNMENT (the frames of its shard plus whatever timing_sync placed on noise), in stream
# order.  gather_psdus wants the slots in local frame order, which costs every rank an index_select over its output set per step.  Here
# the output set travels as it is -- every rank with the same number of rows -- and rank 0, which is handed every rank's frame -> row map
# once, forms the global frame order when somebody asks for it (bench.py: after the clock has stopped).

class SlotBuffers:
    """Receive buffers of gather_slots on rank 0: `world` slabs uint8[rows, slot] of one tensor on a GPU, separate tensors in host memory
    (gloo: see GatherBuffers)."""

    def __init__(self, rows, slot, world, device, dtype=torch.uint8):
        device = torch.device(device)
        self.rows = rows
        if device.type == "cpu":
            self.big = None
            self.parts = [torch.empty((rows, slot), dtype=dtype) for _ in range(world)]
        else:
            self.big = torch.empty((world, rows, slot), dtype=dtype, device=device)
            self.parts = list(self.big.unbind(0))


def gather_maps(perm, rank, world, group=None):
    """perm: int64[frames_per_rank], local frame k -> row of this rank's output set.  Returns int64[world, frames_per_rank] on rank 0 (None
    elsewhere).  One gather, made once per workload.  (World 1 with an initialised process group makes the collective call too.)"""
    if world == 1 and not dist.is_initialized():
        return perm.reshape(1, -1)
    parts = [torch.empty_like(perm) for _ in range(world)] if rank == 0 else None
    dist.gather(perm, parts, dst=0, group=group)
    return torch.stack(parts) if rank == 0 else None


def gather_slots(slots, rank, world, group=None, buffers=None):
    """slots: uint8[rows, slot], this rank's output set as the decode wrote it (the same `rows` on every rank).  One gather; returns on
    rank 0 the list of the ranks' sets (views of `buffers` if given), None elsewhere."""
    if rank == 0:
        parts = buffers.parts if buffers is not None else [torch.empty_like(slots) for _ in range(world)]
    else:
        parts = None
    dist.gather(slots, parts, dst=0, group=group)
    return parts if rank == 0 else None


def order_gathered(parts, perms, n_global):
    """Rank 0: the gathered sets -> uint8[n_global, slot] in GLOBAL frame order (frame r + k * world = row perms[r][k] of rank r's set)."""
    world = len(parts)
    per = perms.shape[1]
    out = torch.empty((per, world) + tuple(parts[0].shape[1:]), dtype=parts[0].dtype, device=parts[0].device)
    for r in range(world):
        out[:, r] = parts[r].index_select(0, perms[r].to(parts[r].device))
    return out.reshape((per * world,) + tuple(parts[0].shape[1:]))[:n_global]
