"""Multi-GPU glue: one process per GPU, tracks sharded round-robin (a spawning track goes
to the rank that owns the fewest live tracks, lowest rank on a tie: round-robin by
creation id while nothing dies, balanced under churn), ONE all-gather of predicted
``bbox_t`` per frame, then replicated association + lifecycle on every rank.

The collective is a plain ``torch.distributed.all_gather_into_tensor`` (backend
"nccl" = RCCL over xGMI on the GPU box; "gloo" in the CPU tests).  The message is
24 B x ceil(max_tracks / world) per rank -- 3 KB at 1024 tracks on 8 GPUs --
latency-bound, so there is exactly one collective per frame and nothing else on the
data path.
"""
from __future__ import annotations

import numpy as np

BBOX_BYTES = 24


def slots_per_rank(max_tracks: int, world: int) -> int:
    """boxes in one rank's segment of the all-gather: no rank ever owns more than ceil(max_tracks / world) tracks (assign_owners)"""
    return max_tracks if world <= 1 else (max_tracks + world - 1) // world


def assign_owners(owners_kept, n_spawn: int, world: int):
    """The replicated ownership rule of the device loop (csrc/dl_lifecycle.h, DLState::owner): every spawning track, in detection order, goes to
    the rank that owns the fewest live tracks at that moment, lowest rank on a tie.  Returns the owners of the n_spawn new tracks."""
    load = [0] * world
    for o in owners_kept:
        load[o] += 1
    out = []
    for _ in range(n_spawn):
        r = min(range(world), key=lambda k: (load[k], k))
        load[r] += 1
        out.append(r)
    return out


def local_segment(owners, live_boxes: np.ndarray, rank: int, world: int, spr: int) -> np.ndarray:
    """What rank `rank` contributes to the all-gather: its own tracks' boxes, in live-list order, padded to spr."""
    seg = np.zeros(spr, live_boxes.dtype)
    j = 0
    for i, o in enumerate(owners):
        if o == rank:
            seg[j] = live_boxes[i]
            j += 1
    assert j <= spr
    return seg


def gathered_to_live_order(gathered: np.ndarray, owners, world: int, spr: int) -> np.ndarray:
    """Inverse mapping executed on device by dl_scatter_kernel (csrc/mot_devloop.hip): live index i reads the segment of its owner at the
    position i has among that rank's tracks."""
    out = np.zeros(len(owners), gathered.dtype)
    cursor = [0] * world
    for i, r in enumerate(owners):
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
