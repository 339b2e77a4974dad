"""world_size-2 CPU test (gloo) of the multi-GPU glue: shard ownership, the single all-gather per frame and
the segment -> live-order mapping that dl_scatter_kernel performs on device."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import mot_amd  # noqa: F401
    from multiple_object_tracking_amd import parallel as par
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(123)                      # same stream on every rank: the replicated live list
    dt = mot_amd.BBOX_DTYPE
    max_tracks = 37
    spr = par.slots_per_rank(max_tracks, world)
    ok = True
    tids = []; owners = []
    next_tid = 0
    for frame in range(40):
        # replicated lifecycle: drop some tracks, spawn some (tids keep growing, order is stable); the owner of a track is part of the
        # replicated list, a spawning track goes to the least loaded rank (par.assign_owners = the rule of csrc/dl_lifecycle.h)
        keep = [rng.integers(0, 10) > 1 for _ in tids]
        tids = [t for t, k in zip(tids, keep) if k]; owners = [o for o, k in zip(owners, keep) if k]
        n_sp = min(int(rng.integers(0, 9)), max_tracks - len(tids))
        owners += par.assign_owners(owners, n_sp, world)
        for _ in range(n_sp):
            tids.append(next_tid); next_tid += 1
        ok &= max([owners.count(r) for r in range(world)] + [0]) <= spr   # no rank ever owns more than a segment holds
        boxes = np.zeros(len(tids), dt)
        for i, t in enumerate(tids):                       # the box every rank WOULD predict for tid t this frame
            boxes[i] = (t * 3 + frame, t + frame, t + frame + 79, t * 3 + frame + 79, t % 3, 0.9)
        # each rank only knows its own shard's predictions
        seg = par.local_segment(owners, boxes, rank, world, spr)
        local = torch.from_numpy(seg.view(np.uint8).copy())
        gathered = par.all_gather_boxes(local).numpy().view(dt)
        full = par.gathered_to_live_order(gathered, owners, world, spr)
        ok &= bool(np.array_equal(full, boxes))
        ok &= spr == (max_tracks + world - 1) // world
    dist.barrier()
    q.put((rank, ok))
    dist.destroy_process_group()


def test_allgather_shard_mapping_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r for r, _ in res) == [0, 1]
    assert all(ok for _, ok in res)
