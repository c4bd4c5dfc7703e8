"""Frame sharding across GPUs (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI).

Frames are independent, so the data path has no collective: global frame i lives on rank i mod G
(BASELINE config 4) and is decoded there; the only exchange is one gather of the decoded PSDU slots to rank 0.
"""
import torch
import torch.distributed as dist


def local_frame_ids(n_global, rank, world):
    """Global indices of the frames rank `rank` decodes (round-robin)."""
    return torch.arange(rank, max(n_global, rank), world)


def gather_psdus(psdu_local, n_global, rank, world, group=None, force_collective=False):
    """psdu_local: uint8[m_local, slot] for this rank's frames in local order (global ids rank, rank+G, ...).
    Returns on rank 0 a uint8[n_global, slot] tensor in GLOBAL frame order, None elsewhere.  One gather.
    force_collective: make the dist.gather call at world 1 too (a one-GPU box then exercises the very RCCL call the
    8-GPU job makes; needs an initialised process group)."""
    if world == 1 and not force_collective:
        return psdu_local
    slot = psdu_local.shape[1]
    m_max = (n_global + world - 1) // world
    pad = psdu_local
    if psdu_local.shape[0] != m_max:
        pad = torch.zeros((m_max, slot), dtype=psdu_local.dtype, device=psdu_local.device)
        pad[:psdu_local.shape[0]] = psdu_local
    parts = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, parts, dst=0, group=group)
    if rank != 0:
        return None
    # parts[r][k] is global frame r + k*world: interleave
    stacked = torch.stack(parts, dim=1).reshape(m_max * world, slot)
    return stacked[:n_global]
