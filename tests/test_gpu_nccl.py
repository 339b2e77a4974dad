"""RCCL leg of the sharded device loop: two ranks on two GPUs, ONE all_gather_into_tensor of the predicted bbox_t per frame
(torch.distributed backend "nccl" = RCCL over xGMI), replicated association, local updates -- both ranks must reproduce
the unsharded oracle.  Needs two visible devices; on the single-GPU test box it skips itself (the driver's 8-GPU node runs it)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import mot_amd
    import orc
    from multiple_object_tracking_amd import parallel as par, synth
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    n, nframes = 96, 6
    scene = synth.Scene(n, 80, stream_id=61, miss_pct=5, fp_pct=3)
    items = list(scene.frames(nframes))
    ctx = mot_amd.MotContext(max_tracks=128, max_dets=128, device=rank, rank=rank, world=world)
    oracle = orc.OracleMot(orc.load_oracle(), 0, 0, 128)
    ok = True
    gathered = None
    for f, (frame, dets) in enumerate(items):
        fd = torch.from_numpy(frame).cuda()
        da = mot_amd.boxes_array(dets) if len(dets) else np.zeros(1, mot_amd.BBOX_DTYPE)
        dd = torch.from_numpy(da.view(np.uint8)).cuda()
        seg_ptr, spr = ctx.step_begin_device(fd.data_ptr())
        local = torch.as_tensor(par.DevArray(seg_ptr, spr * 24), device="cuda")
        if gathered is None:
            gathered = torch.empty(world * spr * 24, dtype=torch.uint8, device="cuda")
        ctx.sync()                                                  # the predict kernel wrote this rank's segment
        par.all_gather_boxes(local, gathered)
        torch.cuda.synchronize()
        ctx.step_finish_device(gathered.data_ptr(), dd.data_ptr(), len(dets))
        ref = oracle.step(frame, dets)
        boxes, tids, _ = ctx.live_tracks()
        ok &= bool(np.array_equal(tids, ref["tids"]))
        ok &= all(np.array_equal(boxes[k], ref["live"][k]) for k in ("l", "t", "b", "r", "type"))
    dist.barrier()
    q.put((rank, ok))
    ctx.close(); oracle.close()
    dist.destroy_process_group()


def test_two_ranks_rccl_all_gather():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL)")
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    assert sorted(r for r, _ in res) == [0, 1] and all(ok for _, ok in res)


def test_bench_sharded_branch_two_ranks_one_gpu():
    """bench.py's sharded branch (begin -> all-gather -> finish, max over ranks, rank-0 JSON line) with two ranks on ONE GPU:
    MOT_BENCH_BACKEND=gloo stages the all-gather through the host, so the code path the driver's N > 1 runs take executes on the
    single-GPU test box too (RCCL itself refuses two ranks on one device).  A smoke test, never a measurement."""
    import json
    import subprocess
    env = dict(os.environ, MOT_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--tracks", "96", "--steady", "0", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["scaling"] == "strong" and j["config"]["live_tracks_end"] == 96
    assert j["smoke_backend"].startswith("gloo") and j["value"] > 0
    # every rank's stage times are in the line (round-4 verdict item 7a): predict / all-gather / chain / residual update, all positive
    rows = j["per_rank_stage_ms"]["ranks"]
    assert [r["rank"] for r in rows] == [0, 1]
    assert all(r[k] > 0 for r in rows for k in ("predict_ms", "gather_ms", "chain_ms")) and all(r["update_ms"] >= 0 for r in rows)


def test_bench_streams_mode_two_ranks_one_gpu():
    """BASELINE configs[4]'s shape -- one independent camera stream per rank, replicas only, no collective (`--mode streams`, weak scaling) -- with
    two ranks on ONE GPU under the launcher, 148 x 148 px templates fed by 120-180 px detections (the multi-scale leg of that configuration).  The
    gloo smoke backend only carries the barrier and the max-over-ranks reduction here: the data path has nothing to exchange.  Round-5 verdict item
    9: the first SCALE run of either mode must not fail on plumbing."""
    import json
    import subprocess
    env = dict(os.environ, MOT_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "streams", "--steps", "4", "--warmup", "2",
           "--tracks", "40", "--size", "148", "--det-sizes", "120", "180", "--steady", "0", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["scaling"] == "weak" and j["value"] > 0
    assert j["config"]["tracks_total"] == 80 and j["config"]["tracks_per_gpu"] == 40 and "no collective" in j["config"]["parallelism"]
    assert "per_rank_stage_ms" not in j or j["per_rank_stage_ms"] is None        # nothing is sharded: no all-gather / replicated-chain rows


@pytest.mark.gpu
def test_bench_eight_ranks_one_gpu():
    """BASELINE configs[3]'s shape -- 1024 tracks sharded 128 per rank over EIGHT ranks -- through bench.py's own launcher on one GPU
    (gloo-staged all-gather: RCCL refuses several ranks on one device).  Smoke test of the 8-process path the driver's SCALE run takes:
    rendezvous, shard ownership over eight ranks, replicated association on eight contexts, one JSON line with eight per-rank rows."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MOT_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "2", "--steady", "0", "--no-cpu-baseline"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["config"]["tracks_per_gpu"] == 128 and j["config"]["live_tracks_end"] == 1024 and j["value"] > 0
    assert [r["rank"] for r in j["per_rank_stage_ms"]["ranks"]] == list(range(8))


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the form the driver uses for N = 1) starts its two ranks itself, as child
    processes of a parent that never touches the GPU, relays rank 0's one JSON line and the exit code.  Two ranks on ONE GPU: the gloo-staged
    smoke backend, as above."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MOT_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--tracks", "96", "--steady", "0",
                          "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["value"] > 0
    # a failing rank must surface as a non-zero exit code of the parent
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--tracks", "96", "--size", "7"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert bad.returncode != 0


class _NcclId(__import__("ctypes").Structure):
    _fields_ = [("internal", __import__("ctypes").c_char * 128)]


def _rccl():
    import ctypes as C
    lib = C.CDLL("librccl.so.1")
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_NcclId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _NcclId, C.c_int]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    return lib


def _native_worker(rank, world, idq, resq):
    import ctypes as C
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import mot_amd
    import orc
    from multiple_object_tracking_amd import synth
    torch.cuda.set_device(rank)
    lib = _rccl()
    uid = _NcclId()
    if rank == 0:
        assert lib.ncclGetUniqueId(C.byref(uid)) == 0
        for _ in range(world - 1):
            idq.put(bytes(uid.internal))
    else:
        uid.internal = idq.get(timeout=120)
    comm = C.c_void_p()
    assert lib.ncclCommInitRank(C.byref(comm), world, uid, rank) == 0
    n, nframes = 96, 6
    scene = synth.Scene(n, 80, stream_id=62, miss_pct=5, fp_pct=3)
    ctx = mot_amd.MotContext(max_tracks=128, max_dets=128, device=rank, rank=rank, world=world)
    oracle = orc.OracleMot(orc.load_oracle(), 0, 0, 128)
    ok = True
    for frame, dets in scene.frames(nframes):
        fd = torch.from_numpy(frame).cuda()
        da = mot_amd.boxes_array(dets) if len(dets) else np.zeros(1, mot_amd.BBOX_DTYPE)
        dd = torch.from_numpy(da.view(np.uint8)).cuda()
        torch.cuda.synchronize()
        ctx.step_frame_sharded(fd.data_ptr(), dd.data_ptr(), len(dets), comm.value)   # predict, ncclAllGather, association, update: one call
        ref = oracle.step(frame, dets)
        boxes, tids, _ = ctx.live_tracks()
        ok &= bool(np.array_equal(tids, ref["tids"]))
        ok &= all(np.array_equal(boxes[k], ref["live"][k]) for k in ("l", "t", "b", "r", "type"))
    ctx.close(); oracle.close()
    lib.ncclCommDestroy(comm)
    resq.put((rank, ok))


def test_native_rccl_entry_point_one_rank():
    """mot_step_frame_sharded on the single-GPU box: a ONE-rank RCCL communicator -- binds librccl at run time, issues the real
    ncclAllGather on the context's stream between predict and association, results equal the oracle"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    idq, resq = ctx.Queue(), ctx.Queue()
    p = ctx.Process(target=_native_worker, args=(0, 1, idq, resq))
    p.start()
    res = resq.get(timeout=600)
    p.join(timeout=120)
    assert res == (0, True)


def test_native_rccl_entry_point_two_ranks():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL)")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    idq, resq = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_native_worker, args=(r, 2, idq, resq)) for r in range(2)]
    for p in procs:
        p.start()
    res = [resq.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
    assert sorted(res) == [(0, True), (1, True)]
