"""Multi-GPU glue: one process per GPU, tracks sharded round-robin by creation id
(``gpu = tid % world``), ONE all-gather of predicted ``bbox_t`` per frame, then
replicated association + lifecycle on every rank.

The collective is a plain ``torch.distributed.all_gather_into_tensor`` (backend
"nccl" = RCCL over xGMI on the GPU box; "gloo" in the CPU tests).  The message is
24 B x max_tracks per rank (ownership tid % world drifts under track churn, so a
segment must hold up to max_tracks boxes) -- 24 KB at 1024 tracks, latency-bound, so
there is exactly one collective per frame and nothing else on the data path.
"""
from __future__ import annotations

import numpy as np

BBOX_BYTES = 24


def slots_per_rank(max_tracks: int, world: int) -> int:
    return max_tracks


def owner_of(tid: int, world: int) -> int:
    return int(tid) % world


def local_segment(live_tids, live_boxes: np.ndarray, rank: int, world: int, spr: int) -> np.ndarray:
    """What rank `rank` contributes to the all-gather: its own tracks' boxes, in live-list order, padded to spr."""
    seg = np.zeros(spr, live_boxes.dtype)
    j = 0
    for i, tid in enumerate(live_tids):
        if owner_of(tid, world) == rank:
            seg[j] = live_boxes[i]
            j += 1
    return seg


def gathered_to_live_order(gathered: np.ndarray, live_tids, world: int, spr: int) -> np.ndarray:
    """Inverse mapping executed on device by dl_scatter_kernel (csrc/mot_devloop.hip): live index i reads
    segment (tid_i % world) at the position i has among that rank's tracks."""
    out = np.zeros(len(live_tids), gathered.dtype)
    cursor = [0] * world
    for i, tid in enumerate(live_tids):
        r = owner_of(tid, world)
        out[i] = gathered[r * spr + cursor[r]]
        cursor[r] += 1
    return out


def all_gather_boxes(local_u8, out_u8=None):
    """the frame's single collective; tensors are uint8 views of bbox_t arrays"""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    if out_u8 is None:
        out_u8 = torch.empty(world * local_u8.numel(), dtype=torch.uint8, device=local_u8.device)
    dist.all_gather_into_tensor(out_u8, local_u8)
    return out_u8


class DevArray:
    """__cuda_array_interface__ view of raw device memory, so torch.distributed can send the context's own
    all-gather segment without a copy."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
