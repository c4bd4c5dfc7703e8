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
