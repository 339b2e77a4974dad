#!/usr/bin/env python3
"""bench.py -- tracker-updates/sec of the MI355X-native per-frame tracker update.

    python bench.py --gpus N --steps K --warmup W

A "step" is one camera frame: every live track runs tracker_predict, the
predicted boxes are (for N > 1) exchanged with ONE RCCL all-gather, the
detections x tracks cost matrix + Munkres assignment run replicated on every
rank, then every track runs tracker_update; track lifecycle (delete / spawn)
included (top/td.cpp:344-644).  One tracker-update = one track x one frame.

Workload (BASELINE.json configs[2] at N=1, configs[3] at N=8): 1024 concurrent
80x80 KCF tracks on a synthetic 1280x720 BGR stream, sharded across the
GPUs of one node => "scaling": "strong".  Frames and detections are resident in
HBM before the timed region.  `--tracks 64` gives configs[1].

Prints ONE JSON line on rank 0 (contract in the task description), including
`roofline` for the dominant kernel (HIP events on the launch stream) and
`cpu_baseline` (the reference built in oracle/_ref, or the oracle port, timed on
one host core on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# algorithmic HBM bytes per track and launch, 80x80 template (SURVEY 8d / DESIGN.md):
#   predict: 19,200 (u8 BGR crop) + 54,560 (xm read) + 880 (alpha) + 48 (pos in, box out)
#   update : 19,200 (crop) + 109,120 (xm read+write) + 1,760 (alpha r+w) + 24 (box in)


def alg_bytes(size):
    nb = (size // 4) * ((size // 4) // 2 + 1)
    crop = size * size * 3
    return {"predict": crop + 31 * nb * 8 + nb * 4 + 48, "update": crop + 2 * 31 * nb * 8 + 2 * nb * 4 + 24,
            # split update (default in the device loop): spectrum of the detection read once, model read + written, alpha r/w
            "blend": 3 * 31 * nb * 8 + 3 * nb * 4 + 24}


def gen_stream(n_tracks, size, n_frames, stream_id=0, det_sizes=None, first_frame_exact=True, miss_pct=0, fp_pct=0, nms=False, counts=None):
    """frames [n_frames][720][1280][3] u8 and detections [n_frames][n_tracks or cap] of a synthetic stream; with misses / false
    positives / nms the number of detections varies per frame: pass a list as `counts` to receive them (rows are zero-padded)"""
    import mot_amd
    from multiple_object_tracking_amd import synth
    scene = synth.Scene(n_tracks, size, stream_id=stream_id, det_sizes=det_sizes, first_frame_exact=det_sizes is not None and first_frame_exact,
                        miss_pct=miss_pct, fp_pct=fp_pct, nms=nms)
    varying = bool(miss_pct or fp_pct or nms)
    cap = min(1024, n_tracks + n_tracks // 8 + 8) if varying else n_tracks
    frames = np.empty((n_frames, 720, 1280, 3), np.uint8)
    dets = np.zeros((n_frames, cap), mot_amd.BBOX_DTYPE)
    for f, (frame, d) in enumerate(scene.frames(n_frames)):
        frames[f] = frame
        d = d[:cap]
        dets[f, :len(d)] = mot_amd.boxes_array(d)
        if counts is not None:
            counts.append(len(d))
    return frames, dets


def _cpu_worker(args, want_state=False):
    """one host core: an independent stream of `n` KCF tracks through the oracle's frame loop; returns (updates, seconds, frames)
    and, with want_state, the live list (boxes, tids) after the last frame -- the parity record the GPU run is checked against"""
    n, size, sid, frames_cap, budget = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    import mot_amd  # noqa: F401
    from multiple_object_tracking_amd import synth
    lib = orc.load_oracle()
    scene = synth.Scene(n, size, stream_id=sid)
    gen = scene.frames(10 ** 6)
    m = orc.OracleMot(lib, 0, 0, max(n, 1))
    frame, dets = next(gen); last = m.step(frame, dets)              # frame 0 spawns the tracks (not steady state)
    t0 = time.perf_counter(); done = 0
    while done < frames_cap and time.perf_counter() - t0 < budget:
        frame, dets = next(gen); last = m.step(frame, dets); done += 1
    dt = time.perf_counter() - t0
    m.close()
    if want_state:
        return n * done, dt, done, (last["live"], last["tids"])
    return n * done, dt, done


def committed_traffic(dom, n_tracks, plain_workload, defer, split, prof_dir=None):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC passes (profiles/rNN_traffic.json), or (None, None).
    The counters were collected on ONE workload -- N tracks of 80 x 80 px on the plain synthetic stream -- and the key carries N only, so any
    other workload (another template size, per-track sizes, detection sizes, detector noise) gets null: round-5 verdict, a 64-96 px per-track
    run carried the 80-px figure of a launch that is not that kernel."""
    if not plain_workload:
        return None, None
    prof_dir = prof_dir or os.path.join(ROOT, "profiles")
    for cand_file in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        tpath = os.path.join(prof_dir, cand_file)
        if not os.path.exists(tpath):
            continue
        try:
            tj = json.load(open(tpath))
            traffic = tj.get(("kcf_predict" if dom.startswith("kcf_predict") else "kcf_update_blend" if split else "kcf_update") + f"_bytes_per_launch_n{n_tracks}")
            if defer and "deferred_blend" not in tj:
                traffic = None                                          # counter passes of a build without the deferred blend: not this kernel
        except Exception:
            traffic = None
        if traffic is None:
            return None, None
        return traffic, (f"profiles/{cand_file}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this command "
                         "(tools/collect_profiles.sh), not measured by this run")
    return None, None


def cpu_baseline(n_tracks, size, budget_s=12.0):
    """CPU baseline on the host cores of this box (rank 0, N = 1 only), bounded samples of the same synthetic workload:
      * value / kind "reference": the reference's own code (oracle/_ref, when present), ONE core (the reference's tracker thread is
        single-threaded, top/td.cpp:306), full track count, all stages incl. its Munkres;
      * port_1core: the oracle port on one core, same frames -> port_over_reference = calibration ratio;
      * all_cores: the port on every host core, tracks split into one independent stream per core (n / cores tracks each, each with
        its own cost matrix + Munkres) -- an upper bound for a multi-threaded CPU build, count stated."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    import mot_amd  # noqa: F401
    from multiple_object_tracking_amd import synth
    out = {}
    libs = None
    if orc.ref_available():
        try:
            libs = tuple(orc.load_ref(n) for n in ["kcf", "kalman", "hungarian", "drawlib"])
        except OSError:
            libs = None
    # oracle port, one core, the first steady frames of the bench stream
    upd_p, t_p, nf_p, state_p = _cpu_worker((n_tracks, size, 0, 3, budget_s * 0.4), want_state=True)
    port = upd_p / t_p
    out["_parity"] = {"frames": nf_p, "port": state_p}                # popped by main(): what the GPU must reproduce after frame nf_p
    if libs is not None:
        class Feed:
            def __init__(self, items): self.items = items
            def frames(self, n): return iter(self.items[:n])
        scene = synth.Scene(n_tracks, size, stream_id=0)
        gen = scene.frames(10 ** 6)
        nf = max(1, nf_p)
        frames = [next(gen) for _ in range(nf + 1)]
        timing = []
        trace = orc.ref_frame_loop(0, Feed(frames), nf + 1, libs, timing)
        out["_parity"]["reference"] = (mot_amd.boxes_array([tuple(b) + (0.9,) for b in trace[-1]["live"]]), np.array(trace[-1]["tids"], np.uint32))
        t_used = sum(t for t, _ in timing[1:]); updates = sum(n for _, n in timing[1:])
        out.update({"value": updates / t_used, "unit": "tracker-updates/s", "cores": 1, "kind": "reference",
                    "sample": f"{n_tracks} KCF tracks x frames 1..{nf} of the bench stream (the spawn frame 0 excluded; the GPU window starts at frame 1 + warmup: same stream, later frames) -- crop+resize, predict, cost, Munkres, update, 1 thread",
                    "port_1core": port, "port_over_reference": port / (updates / t_used)})
    else:
        out.update({"value": port, "unit": "tracker-updates/s", "cores": 1, "kind": "port",
                    "sample": f"{n_tracks} KCF tracks x {nf_p} frames after the spawn frame of the bench stream, oracle port, 1 thread"})
    # all host cores: one independent stream of n / cores tracks per core
    try:
        import multiprocessing as mp
        cores = os.cpu_count() or 1
        per = max(n_tracks // cores, 1)
        with mp.get_context("spawn").Pool(cores) as pool:
            res = pool.map(_cpu_worker, [(per, size, 100 + i, 10 ** 6, budget_s * 0.4) for i in range(cores)])
        out["all_cores"] = {"value": sum(u for u, _, _ in res) / max(t for _, t, _ in res), "cores": cores, "kind": "port",
                            "sample": f"{cores} independent streams of {per} KCF tracks (one per core), {min(f for _, _, f in res)}-{max(f for _, _, f in res)} frames each"}
    except Exception as e:                                            # never fail the bench line over the optional leg
        out["all_cores"] = {"error": str(e)[:200]}
    return out


def dropin_timing(device, n=16, reps=12):
    """us per call of the reference's per-object interface (td.cpp:229-232) through libmot_dropin_kcf.so -- the literal drop-in
    mode: every call is a batch of ONE -- next to the reference's own tracker_predict / tracker_update (kcf.cpp:455-476) on one host
    core, same patches.  Calls are issued in td.cpp's order: per frame, tracker_predict for every object (td.cpp:344-384), then
    tracker_update for every object (td.cpp:512-582).  `per_object_frame_us` is the wall time of a whole frame's calls / objects (it
    contains every wait, whichever call it falls into)."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    import mot_amd
    os.environ.setdefault("MOT_DEVICE", str(device))
    lib = C.CDLL(mot_amd.DROPIN_KCF_PATH)
    new = getattr(lib, "_Z11tracker_newP11_bbox_pos_s"); new.restype = C.c_void_p; new.argtypes = [C.c_void_p]
    pred = getattr(lib, "_Z15tracker_predictPvPfP11_bbox_pos_s"); pred.argtypes = [C.c_void_p] * 3
    upd = getattr(lib, "_Z14tracker_updatePvPfP11_bbox_pos_s"); upd.argtypes = [C.c_void_p] * 3
    dele = getattr(lib, "_Z14tracker_deletePv"); dele.argtypes = [C.c_void_p]
    rng = np.random.default_rng(7)
    patches = [np.ascontiguousarray(rng.integers(0, 256, 6400).astype(np.float32)) for _ in range(n)]
    boxes = [mot_amd.BBox(100 + 5 * i, 100, 179, 179 + 5 * i, i % 3, 0.9) for i in range(n)]

    def run(new_f, pred_f, upd_f, del_f):
        hs = [C.c_void_p(new_f(C.byref(b))) for b in boxes]
        for h, p, b in zip(hs, patches, boxes):
            upd_f(h, orc.P(p), C.byref(b))                              # first update (eta = 1), td.cpp:631-640
        pb = mot_amd.BBox()
        pred_f(hs[0], orc.P(patches[0]), C.byref(pb))                   # (drains whatever the first updates left queued)
        tp, tu = [], []
        t_all = time.perf_counter()
        for _ in range(reps):
            for h, p in zip(hs, patches):
                t0 = time.perf_counter(); pred_f(h, orc.P(p), C.byref(pb)); tp.append((time.perf_counter() - t0) * 1e6)
            for h, p, b in zip(hs, patches, boxes):
                t0 = time.perf_counter(); upd_f(h, orc.P(p), C.byref(b)); tu.append((time.perf_counter() - t0) * 1e6)
        pred_f(hs[0], orc.P(patches[0]), C.byref(pb))                   # the last updates have run when this returns
        t_all = (time.perf_counter() - t_all) * 1e6
        for h in hs:
            del_f(h)
        return np.array(tp), np.array(tu), t_all / (reps * n)
    run(new, pred, upd, dele)                                          # warm-up (context creation, first launches)
    p_us, u_us, frame_us = run(new, pred, upd, dele)
    out = {"tracker_predict_us": float(p_us.mean()), "tracker_update_us": float(u_us.mean()),
           "tracker_predict_p50_p90_us": [float(np.percentile(p_us, 50)), float(np.percentile(p_us, 90))],
           "tracker_update_p50_p90_us": [float(np.percentile(u_us, 50)), float(np.percentile(u_us, 90))],
           "per_object_frame_us": float(frame_us),
           "calls": n * reps, "patch": "80x80 float gray (caller memory)", "order": "td.cpp: all predicts of a frame, then all updates",
           "note": "per-object interface = batch of one per call, zero-copy (the kernel reads the pinned patch and writes the box itself): tracker_predict = one launch + one stream "
                   "synchronisation; tracker_update returns with its launch queued (two staging halves), so back-to-back updates run at the kernel's rate and the first predict of the "
                   "next frame waits for the last one; the batch ABI (mot_step_frame_device) is the measured path"}
    if orc.ref_available():
        k = orc.load_ref("kcf")
        rp, ru, rf = run(lambda b: k.refkcf_new(b), k.refkcf_predict, k.refkcf_update, k.refkcf_delete)
        out.update({"reference_tracker_predict_us": float(rp.mean()), "reference_tracker_update_us": float(ru.mean()), "reference_per_object_frame_us": float(rf), "reference_cores": 1})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--tracks", type=int, default=1024, help="total concurrent KCF tracks (all GPUs)")
    ap.add_argument("--size", type=int, default=80, help="square template / object size in pixels")
    ap.add_argument("--det-sizes", type=int, nargs=2, default=None, metavar=("LO", "HI"),
                    help="multi-scale detections (BASELINE configs[4]): tracks are spawned with the --size template from frame 0, "
                         "from frame 1 on every detection is a random square of LO..HI px that the tracker resizes to its template")
    ap.add_argument("--per-track-sizes", action="store_true",
                    help="with --det-sizes LO HI: every track keeps the template size of the detection that spawned it (the reference's "
                         "behaviour, kcf.cpp:148-152) -- one pool per size LO..HI in the device-resident loop; informative, not the headline config")
    ap.add_argument("--miss-pct", type=int, default=0, help="detector noise: every object is missed with this probability (percent) in every frame")
    ap.add_argument("--fp-pct", type=int, default=0, help="detector noise: number of false-positive boxes per frame, percent of --tracks")
    ap.add_argument("--nms", action="store_true", help="no two detections of a frame share a centroid (what a detector's NMS guarantees)")
    ap.add_argument("--mode", choices=["sharded", "streams"], default="sharded",
                    help="N > 1: sharded = ONE stream of --tracks tracks, tracks sharded over the N ranks (least-loaded rank per spawn), one all-gather per frame (BASELINE configs[3], strong scaling); "
                         "streams = N independent camera streams of --tracks tracks each, one per GPU, no collective (BASELINE configs[4], weak scaling)")
    ap.add_argument("--streams-per-gpu", type=int, default=1,
                    help="single GPU only: K independent camera streams of --tracks tracks each run concurrently (K contexts, K HIP streams); "
                         "value = all K streams. Informative (small track counts leave most CUs idle); the default 1 is the measured config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-frames", type=int, default=20)
    ap.add_argument("--steady", type=int, default=40, help="frames of a second timed window right behind the first (reported as steady_state; 0 = off)")
    ap.add_argument("--h2d", type=int, default=1,
                    help="1: a second context runs the SAME frames as the timed window with every frame (2.76 MB, pinned host memory) and its "
                         "detection list uploaded inside the timed region, on a copy stream, double-buffered against the previous frame "
                         "(td.cpp:326-333: the tracker thread receives each frame from the capture side); reported as h2d_inclusive, never as value; 0 = off")
    ap.add_argument("--no-dropin", action="store_true", help="skip timing the per-object drop-in interface (tracker_predict / tracker_update through libmot_dropin_kcf.so)")
    ap.add_argument("--debug-assoc", action="store_true", help="print Munkres step counters / phase times per profiled frame to stderr")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as CHILD processes of a parent that has not touched
    # the GPU (no torch import, no HIP call -- a process that has initialised the GPU must never exec or be replaced), relay rank 0's
    # JSON line and hand the children's exit code on.  Under torch.distributed.run (WORLD_SIZE set) this branch is skipped.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MOT_BENCH_SELF_LAUNCHED="1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
        for ln in proc.stdout.splitlines():
            if ln.startswith("{"):
                print(ln, flush=True)
        raise SystemExit(proc.returncode)

    import torch
    import torch.distributed as dist
    import mot_amd
    from multiple_object_tracking_amd import parallel as par

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: launch it as `python bench.py --gpus N` or under "
                         "`python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # MOT_BENCH_BACKEND=gloo is a smoke-test mode for boxes with fewer GPUs than ranks: every rank uses cuda:0 and the
    # all-gather is staged through the host.  It exercises the sharded device path, not RCCL, and is never a measurement.
    backend = os.environ.get("MOT_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    red_dev = "cpu" if backend == "gloo" else "cuda"                 # where the scalar reductions of the timing live
    n_tracks, size = args.tracks, args.size
    streams = args.mode == "streams" and world > 1
    mot_rank, mot_world = (0, 1) if streams else (rank, world)
    n_prof = args.profile_frames if (world == 1 and args.streams_per_gpu == 1) else 0
    n_h2d = args.h2d if (world == 1 and args.streams_per_gpu == 1) else 0
    n_inloop = 20 if n_prof else 0                                      # frames of the ordinary loop whose predict launch is timed in place (roofline)
    n_rank_prof = 10 if (world > 1 and not streams) else 0            # sharded runs: frames whose per-rank stage times are recorded (behind the timed windows)
    n_frames = 1 + args.warmup + args.steps + args.steady + n_inloop + n_prof + n_rank_prof
    det_counts = []
    frames_h, dets_h = gen_stream(n_tracks, size, n_frames, stream_id=rank if streams else 0, det_sizes=tuple(args.det_sizes) if args.det_sizes else None,
                                  first_frame_exact=not args.per_track_sizes, miss_pct=args.miss_pct, fp_pct=args.fp_pct, nms=args.nms, counts=det_counts)
    cap_t = min(1024, max(n_tracks, dets_h.shape[1]))                   # tracks come and go with detector noise
    dev_sizes = tuple(args.det_sizes) if (args.per_track_sizes and args.det_sizes) else None
    frames_d = torch.from_numpy(frames_h).cuda()
    dets_d = torch.from_numpy(dets_h.view(np.uint8).reshape(n_frames, -1)).cuda()
    frame_bytes = 720 * 1280 * 3
    det_bytes = dets_d.shape[1]

    stream = torch.cuda.Stream()
    ctx = mot_amd.MotContext(tracker_kind=mot_amd.TRACKER_KCF, device=local_rank, max_tracks=max(cap_t, 1), max_dets=max(cap_t, 1),
                             rank=mot_rank, world=mot_world, stream=stream.cuda_stream, dev_size=size, dev_sizes=dev_sizes)

    gathered = None
    extra = []                                                       # --streams-per-gpu: more independent contexts on streams of their own
    if args.streams_per_gpu > 1 and world == 1:
        for _ in range(args.streams_per_gpu - 1):
            st = torch.cuda.Stream()
            extra.append((st, mot_amd.MotContext(tracker_kind=mot_amd.TRACKER_KCF, device=local_rank, max_tracks=max(n_tracks, 1),
                                                 max_dets=max(n_tracks, 1), stream=st.cuda_stream, dev_size=size, dev_sizes=dev_sizes)))

    def step(f):
        fp = frames_d.data_ptr() + f * frame_bytes
        dp = dets_d.data_ptr() + f * det_bytes
        if mot_world == 1:
            # one frame of look-ahead: the next frame and its detection list are resident too (as they are for the reference's tracker
            # thread, which runs behind the detector through a 64-slot ring), so the library may compute the NEXT frame's detection
            # features beside this frame's association chain.  Same results (tests); MOT_LOOKAHEAD=0 computes everything in-frame.
            nf, nd, nn = (fp + frame_bytes, dp + det_bytes, det_counts[f + 1]) if f + 1 < n_frames else (0, 0, 0)
            ctx.step_frame_device_ahead(fp, dp, det_counts[f], nf, nd, nn)
            for _, cx in extra:
                cx.step_frame_device_ahead(fp, dp, det_counts[f], nf, nd, nn)
        else:
            nf, nd, nn = (fp + frame_bytes, dp + det_bytes, det_counts[f + 1]) if f + 1 < n_frames else (0, 0, 0)
            seg_ptr, spr = ctx.step_begin_device_ahead(fp, dp, det_counts[f], nf, nd, nn)   # look-ahead of the detection features, as on one GPU
            nonlocal gathered
            if gathered is None:
                gathered = torch.empty(mot_world * spr * 24, dtype=torch.uint8, device="cuda")
                step.local = torch.as_tensor(par.DevArray(seg_ptr, spr * 24), device="cuda")
            if backend == "gloo":
                stream.synchronize()
                gathered.copy_(par.all_gather_boxes(step.local.cpu()))
            else:
                par.all_gather_boxes(step.local, gathered)         # the single collective of the frame (RCCL over xGMI)
            ctx.step_finish_device(gathered.data_ptr(), dp, det_counts[f])

    with torch.cuda.stream(stream):
        f = 0
        step(f); f += 1                                              # frame 0: every detection spawns a track (tracker_new + first update)
        for _ in range(args.warmup):
            step(f); f += 1
        stream.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(f); f += 1
        stream.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        elapsed = t1 - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        n_live = ctx.live_count() + sum(cx.live_count() for _, cx in extra)
        if streams:                                                  # every rank tracks its own stream: whole-job units = sum over ranks
            t = torch.tensor([n_live], dtype=torch.int64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            n_live = int(t.item())

        # second timed window right behind the first (same loop, no events): the stream has left its start-up transient
        steady = None
        if args.steady > 0:
            stream.synchronize(); torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            ts0 = time.perf_counter()
            for _ in range(args.steady):
                step(f); f += 1
            stream.synchronize(); torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            ts = time.perf_counter() - ts0
            if world > 1:
                t = torch.tensor([ts], dtype=torch.float64, device=red_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                ts = float(t.item())
            steady = {"value": n_live * args.steady / ts, "ms_per_step": ts / args.steady * 1e3, "frames": args.steady,
                      "first_frame": 1 + args.warmup + args.steps}

        # sharded: what a frame costs THIS rank, stage by stage (HIP events on the context's stream around predict launch / all-gather /
        # replicated association chain / residual update), mean over 10 ordinary frames behind the timed windows; all ranks' rows go into
        # the line so a scaling run can be checked against DESIGN.md section 5's table
        rank_stages = None
        if n_rank_prof:
            # an informative leg: it must never cost the line -- and never leave the other ranks alone in a collective (round-5 advisor finding):
            # every rank ALWAYS executes its n_rank_prof frames (each holds the frame's all-gather); only the profile calls may fail
            acc4 = np.zeros(4); perr = None
            for _ in range(n_rank_prof):
                try: ctx.debug_profile_stages(True)
                except Exception as e: perr = perr or str(e)[:200]
                step(f); f += 1
                try: acc4 += ctx.debug_profile_stages(False, read=True)
                except Exception as e: perr = perr or str(e)[:200]
            if perr is None: mine = {"rank": rank, **{k: float(v) for k, v in zip(("predict_ms", "gather_ms", "chain_ms", "update_ms"), acc4 / n_rank_prof)}}
            else: mine = {"rank": rank, "error": perr}
            rows = [None] * world
            try:
                dist.all_gather_object(rows, mine)
                rank_stages = rows
            except Exception as e:
                rank_stages = [{"rank": rank, "error": "all_gather_object: " + str(e)[:200]}]

        # the predict launch as it runs IN the loop (look-ahead feature launch beside it, deferred blend in its prologue): its own begin / end
        # stamps over 20 ordinary frames -- the duration rocprofv3 --kernel-trace reports for it in this configuration (roofline.avg_launch_ms)
        inloop_ms = None
        if n_inloop:
            ctx.debug_predict_timing(n_inloop)
            for _ in range(n_inloop):
                step(f); f += 1
            tm = ctx.debug_predict_times(n_inloop)
            ctx.debug_predict_timing(0)
            if len(tm):
                inloop_ms = float(np.mean(tm))

        # per-kernel device time, HIP events on the launch stream (world == 1 only)
        stage = None
        assoc_ms, used_by = [], [0, 0, 0]
        if n_prof:
            acc = np.zeros(5)
            for _ in range(n_prof):
                fp = frames_d.data_ptr() + f * frame_bytes
                dp = dets_d.data_ptr() + f * det_bytes
                st5 = ctx.profile_frame_device(fp, dp, det_counts[f])
                acc += st5
                assoc_ms.append(float(st5[1] + st5[3]))
                used_by[int(ctx.lap_stats()[15]) % 3] += 1
                if args.debug_assoc:
                    print("assoc", ctx.assoc_stats().tolist(), file=sys.stderr)
                    ls = ctx.lap_stats()
                    print(f"lap frame {f}: chain {assoc_ms[-1] * 1e3:.0f} us | cert outcome {ls[0]} solver rounds {ls[1]} free {ls[2]} ticks {ls[7]} | "
                          f"sparse status {ls[8]} aug {ls[9]} s5 {ls[10]} events {ls[11]} t_s3 {ls[12] / 100:.1f} us t_s5 {ls[13] / 100:.1f} us total {ls[14] / 100:.1f} us | decided {ls[15]}", file=sys.stderr)
                    pa, ub = ctx.debug_kcf_phases(True)
                    print("kcf predict phases us", (np.diff(pa) / 100.0).round(1).tolist(), "update", (np.diff(ub) / 100.0).round(1).tolist(), file=sys.stderr)
                f += 1
            stage = acc / n_prof
        # round 6: tie frames whose certificate fails only on a few disjoint two-row cycles are committed at once; the emulation (on a stream of its own,
        # beside the next predict) only names the swapped pairs.  Cumulative counters of this context over everything it has stepped so far.
        prov_stats = None
        if world == 1:
            ls_ = ctx.lap_stats()
            prov_stats = {"tie_frames": int(ls_[20]), "committed_provisionally": int(ls_[21]), "pairs_swapped": int(ls_[22]), "bits_by_dense_emulation": int(ls_[23]),
                          "certified_frames": int(ls_[16]), "switch": os.environ.get("MOT_PROV", "1"),
                          "note": "in the timed windows the emulation of a provisionally committed frame runs beside the next frame's predict launch; the profile frames behind "
                                  "them (ms_* above) are synchronised per frame, so their chain time includes the whole emulation as it did in rounds 3-5"}

        # third timed window: the frame and its detections arrive from pinned host memory INSIDE the timed region (SURVEY 8d:
        # "frame-level costs (H2D of the frame ...) are included in wall time").  Copy stream + three device buffers: the upload of
        # frame f + 2 overlaps the association chain of frame f; a buffer is reused only after the frame that read it has finished.
        h2d = None
        if n_h2d:
            # a FRESH context over the SAME frames as `value`: frames 0 .. warmup host-fed and untimed, then exactly `steps` frames timed
            # (round-3 verdict: the two numbers must be comparable frame for frame)
            n_host = 1 + args.warmup + args.steps
            pin_f = torch.from_numpy(frames_h[:n_host]).pin_memory()
            pin_d = torch.from_numpy(dets_h[:n_host].view(np.uint8).reshape(n_host, -1)).pin_memory()
            scratch = torch.empty_like(pin_f, device="cuda"); scratch.copy_(pin_f, non_blocking=False); del scratch   # first DMA touch of the freshly pinned pages: not part of a steady stream
            hctx = mot_amd.MotContext(tracker_kind=mot_amd.TRACKER_KCF, device=local_rank, max_tracks=max(cap_t, 1), max_dets=max(cap_t, 1),
                                      stream=stream.cuda_stream, dev_size=size, dev_sizes=dev_sizes)
            for k in range(1 + args.warmup):
                hctx.step_frame_host(pin_f[k].data_ptr(), pin_d[k].data_ptr(), det_counts[k])
            stream.synchronize(); torch.cuda.synchronize()
            th0 = time.perf_counter()
            for k in range(1 + args.warmup, n_host):                    # the library uploads on its own copy stream, two device buffers
                hctx.step_frame_host(pin_f[k].data_ptr(), pin_d[k].data_ptr(), det_counts[k])
            stream.synchronize(); torch.cuda.synchronize()
            th = time.perf_counter() - th0
            h2d = {"h2d": "included", "value": hctx.live_count() * args.steps / th, "unit": "tracker-updates/s", "ms_per_step": th / args.steps * 1e3, "frames": args.steps,
                   "first_frame": 1 + args.warmup, "same_frames_as_value": True, "bytes_per_frame": frame_bytes + det_counts[1 + args.warmup] * 24,
                   "how": "a fresh context fed through mot_step_frame_host over the same frames as `value`: pinned host frames, copy kernel on the context's copy stream, three device buffers: the upload of frame f + 2 and the detection features of frame f + 1 run beside the association chain of frame f"}
            hctx.close()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_live * args.steps / elapsed
        out = {
            "metric": "tracker-updates/sec (KCF, 80x80 patch)", "value": value, "unit": "tracker-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            **({"rccl_ranks": world, "collective": "1 x all_gather_into_tensor(bbox_t[ceil(max_tracks / world)] per rank) per frame" if not streams else "none"} if world > 1 and backend != "gloo" else {}),
            **({"smoke_backend": "gloo (not a measurement)"} if backend == "gloo" and world > 1 else {}),
            "higher_is_better": True, "scaling": "weak" if streams else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{world} independent camera streams (one per GPU, no collective), each {n_tracks} concurrent {size}x{size} KCF tracks, "
                                    f"1280x720 BGR synthetic stream, Munkres {n_tracks}x{n_tracks}; BASELINE configs[4]") if streams else
                                   (f"{n_tracks} concurrent {size}x{size} KCF tracks (31-ch FHOG, cell 4), 1280x720 BGR synthetic stream, "
                                    f"{n_tracks} detections/frame, Munkres {n_tracks}x{n_tracks}, tracks sharded over {world} rank(s); "
                                    f"BASELINE configs[{2 if n_tracks == 1024 else 1}]" + ("/[3]" if world > 1 else "")),
                       **({"streams_per_gpu": args.streams_per_gpu, "note": "K independent contexts on K HIP streams of one GPU; value = all streams"} if extra else {}),
                       "tracks_total": n_tracks * (world if streams else 1) * (1 + len(extra)), "tracks_per_gpu": n_tracks if streams else n_tracks // world,
                       "live_tracks_end": n_live, "patch": size, **({"det_sizes": args.det_sizes} if args.det_sizes else {}),
                       **({"detector_noise": {"miss_pct": args.miss_pct, "fp_pct": args.fp_pct, "nms": bool(args.nms)}} if (args.miss_pct or args.fp_pct or args.nms) else {}),
                       **({"per_track_template_sizes": True} if dev_sizes else {}),
                       "parallelism": (f"{world} replicas, no collective" if streams else f"track-shard x{world}, 1 all-gather/frame") if world > 1 else "single GPU"},
        }
        ab = alg_bytes(size)
        if steady is not None:
            out["steady_state"] = steady
        if rank_stages is not None:
            out["per_rank_stage_ms"] = {"frames": n_rank_prof, "first_frame": 1 + args.warmup + args.steps + args.steady, "ranks": rank_stages,
                                        "what": "mean per frame on every rank: predict launch (own shard), all-gather (end of the predict -> start of the finish call), replicated association chain incl. scatter + lifecycle, residual update launch; HIP events on the context's stream"}
        if h2d is not None:
            out["h2d_inclusive"] = h2d
        if stage is not None:
            # the lifecycle step is the tail of the final association kernel; stage[3] is only the gap between two event records
            kern = {"kcf_predict": stage[0], "association (row scan, LAP solver, dual check, sparse / dense Munkres, lifecycle)": stage[1] + stage[3], "kcf_update": stage[4]}
            split = os.environ.get("MOT_SPLIT_UPDATE", "1") != "0" and not dev_sizes
            # deferred blend (default with the split update): the model update of frame f (kf, alpha, model lerp: everything of
            # tracker_update but its feature half) runs as the prologue of frame f + 1's predict kernel -- that launch then moves the
            # predict's bytes plus the update's model bytes; what is left on the main stream behind the association is a small launch
            # for the tracks that keep their predicted box
            defer = split and os.environ.get("MOT_DEFER_BLEND", "1") != "0"
            # roofline: the HBM-bound kernel with the largest device time.  With the split update the update stage on the main stream is
            # the blend launch (its feature half runs beside the association on the side stream)
            if defer:
                crop = size * size * 3
                cands = {"kcf_predict (+ deferred model update of the previous frame)": (stage[0], ab["predict"] + ab["update"] - crop),
                         "kcf_update (tracks that keep their predicted box; none on this stream)": (max(stage[4], 1e-6), 0)}
            else:
                cands = {"kcf_predict": (stage[0], ab["predict"]), ("kcf_update (blend launch)" if split else "kcf_update"): (stage[4], ab["blend"] if split else ab["update"])}
            dom = max(cands, key=lambda k: cands[k][0])
            per_launch = cands[dom][1] * n_live
            isolated_ms = cands[dom][0]
            # the line's `frac` uses the launch AS IT RUNS IN THE TIMED LOOP (overlapping the look-ahead feature launch): the duration a
            # rocprofv3 --kernel-trace of this command reports; the isolated launch of the profile frames is given beside it
            launch_ms = inloop_ms if (inloop_ms is not None and dom.startswith("kcf_predict")) else isolated_ms
            achieved = per_launch / (launch_ms * 1e-3) / 1e9
            plain = size == 80 and not (args.det_sizes or args.per_track_sizes or args.miss_pct or args.fp_pct or args.nms)
            traffic, traffic_src = committed_traffic(dom, n_tracks, plain, defer, split)
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                               "alg_bytes_per_launch": per_launch, "avg_launch_ms": launch_ms,
                               "measured": "in the loop" if launch_ms is not isolated_ms else "isolated (profile frames)",
                               "isolated": {"avg_launch_ms": isolated_ms, "frac": per_launch / (isolated_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                            "what": "the same launch in the profile frames: no look-ahead, nothing else on the chip"},
                               "note": "HIP events holding the kernel's own begin / end time stamps (hipExtLaunchKernelGGL start / stop events on the launch stream) over 20 ordinary frames of the loop: the duration rocprofv3 --kernel-trace reports for the launch in the timed configuration, without the marker packets of an event pair around it"}
            other = [k for k in cands if k != dom][0]
            ob = cands[other][1] * n_live
            out["roofline_other"] = {"bound": "hbm", "kernel": other, "achieved": ob / (cands[other][0] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": ob / (cands[other][0] * 1e-3) / 1e9 / HBM_PEAK_GBS, "alg_bytes_per_launch": ob, "avg_launch_ms": cands[other][0]}
            # the association stage moves almost no data: it is reported as latency-bound with its share of the frame, not against a roofline
            tot = stage[0] + stage[1] + stage[3] + stage[4]
            am = np.array(assoc_ms)
            out["latency_bound"] = {"kernel": "association chain (lap_rowscan, lap_solve, lap_verify, mk_sparse, mk_postcheck, munkres/lifecycle)",
                                    "share_of_frame": float((stage[1] + stage[3]) / tot), "ms_mean": float(am.mean()), "ms_p50": float(np.percentile(am, 50)),
                                    "ms_p90": float(np.percentile(am, 90)), "ms_max": float(am.max()),
                                    "decided_by": {"certificate": used_by[0], "sparse_emulation": used_by[1], "dense_emulation": used_by[2]},
                                    "provisional_commits": prov_stats,
                                    "frames": len(assoc_ms), "first_frame": 1 + args.warmup + args.steps + args.steady + n_inloop}
            out["kernel_ms"] = {k: float(v) for k, v in kern.items()}
            out["hbm_frac_whole_frame"] = (ab["predict"] + ab["update"]) * n_live / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS
        parity_fail = False
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(n_tracks, size)
            par_rec = cb.pop("_parity", None)
            default_stream = not (args.det_sizes or args.miss_pct or args.fp_pct or args.nms or args.per_track_sizes)
            if par_rec and default_stream and par_rec["frames"] >= 1:
                # parity check with every measurement: a fresh context runs frames 0..K of the SAME stream through the
                # device-resident loop; its live list (boxes, types, track ids) must equal what the CPU legs just computed
                K = par_rec["frames"]
                vctx = mot_amd.MotContext(tracker_kind=mot_amd.TRACKER_KCF, device=local_rank, max_tracks=max(cap_t, 1), max_dets=max(cap_t, 1),
                                          stream=stream.cuda_stream, dev_size=size)
                with torch.cuda.stream(stream):
                    for k in range(K + 1):
                        vctx.step_frame_device(frames_d.data_ptr() + k * frame_bytes, dets_d.data_ptr() + k * det_bytes, det_counts[k])
                    vb, vt, _ = vctx.live_tracks()
                vctx.close()
                res = {}
                for name in ("port", "reference"):
                    if name in par_rec:
                        eb, et = par_rec[name]
                        res[name] = bool(len(eb) == len(vb) and np.array_equal(np.asarray(et, np.uint32), vt)
                                         and all(np.array_equal(eb[k2], vb[k2]) for k2 in ("l", "t", "b", "r", "type")))
                out["parity_checked"] = {"ok": all(res.values()), "frames": f"0..{K}", "tracks": int(len(vb)), "equal_to": res,
                                         "what": "live boxes (l,t,b,r,type) and track ids of the device-resident loop after frame K vs the oracle port"
                                                 + (" and the reference-built loop" if "reference" in res else "") + " of the cpu_baseline leg, bit-equal"}
                parity_fail = not out["parity_checked"]["ok"]
            else:
                out["parity_checked"] = None
            if not args.no_dropin and size == 80:
                try:
                    cb["dropin"] = dropin_timing(local_rank)
                except Exception as e:                                  # informative leg: never fail the line over it
                    cb["dropin"] = {"error": str(e)[:200]}
            out["cpu_baseline"] = cb
        print(json.dumps(out))
        if parity_fail:
            print("bench.py: PARITY CHECK FAILED -- the number above is not valid", file=sys.stderr)
            ctx.close()
            raise SystemExit(3)
    for _, cx in extra:
        cx.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
