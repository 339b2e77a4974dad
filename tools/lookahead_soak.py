#!/usr/bin/env python3
"""Soak of the look-ahead entry point bench.py times (mot_step_frame_device_ahead), GPU box only.

Repeats the scenario of tests/test_gpu_devloop.py::test_device_loop_lookahead_vs_oracle -- nine frames, a plain call at frame 3, a
deliberately WRONG announcement at frame 5 -- and compares the live list of every frame (or only the last one: --sparse-checks, nothing
synchronised in between) with the oracle's.  Between the calls the host waits a random 0-200 us, and a second context on the same GPU keeps
the chip busy with its own 1024-track stream (--hammer).  The MOT_* switches are read once per process, so one process = one variant;
tools/lookahead_soak_matrix.sh runs the matrix the round-4 verdict asked for.  A mismatch is written to --dump as an .npz (scene parameters,
the repetition, the frame, both live lists, the association statistics, the response maps of the differing tracks): the scene is seeded, so
`lookahead_soak.py N MISS FP 1` under the same environment replays the inputs exactly.

--state: additionally the device state of every (--state-stride-th) live track -- model xm, alpha, pos, scale, pending detection, response map --
is read back at every check point and compared BIT BY BIT with the first repetition's (GPU against GPU: nothing in the path is allowed to depend
on timing, so any difference is a bug even when no box moves; a wrong model only shows in the boxes when its track goes unmatched again).

usage: lookahead_soak.py N MISS FP REPS [--hammer] [--sparse-checks] [--dirty] [--state] [--dump DIR] [--wrong-at F]"""
import argparse, json, os, random, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import mot_amd, orc
from multiple_object_tracking_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("n", type=int); ap.add_argument("miss", type=int); ap.add_argument("fp", type=int); ap.add_argument("reps", type=int)
ap.add_argument("--hammer", action="store_true"); ap.add_argument("--sparse-checks", action="store_true"); ap.add_argument("--dirty", action="store_true")
ap.add_argument("--dump", default=os.path.join(ROOT, "gpurun_out", "soak")); ap.add_argument("--wrong-at", type=int, default=5)
ap.add_argument("--cap", type=int, default=1024); ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--state", action="store_true"); ap.add_argument("--state-stride", type=int, default=1)
ap.add_argument("--frames", type=int, default=9, help="frames per repetition (default 9; with --sparse-checks --state a short run compares the complete device state at its end)")
ap.add_argument("--trace", action="store_true", help="MOT_TRACE=1: the predict / update launches leave one record per workgroup (32 bytes); a wrong run is compared with the last good one record by record")
ap.add_argument("--snap", action="store_true", help="record a stream-ordered device snapshot behind every frame (mot_debug_snapshot, no host synchronisation) and, when the run ends wrong, report frame by frame where it left the oracle")
a = ap.parse_args()
if a.trace:
    os.environ["MOT_TRACE"] = "1"
rng = random.Random(a.seed)
KEYS = ("l", "t", "b", "r", "type")


def dev(frames, dets):
    fd = torch.from_numpy(np.stack(frames)).cuda()
    nmax = max(len(d) for d in dets)
    da = np.zeros((len(dets), max(nmax, 1)), mot_amd.BBOX_DTYPE)
    for i, d in enumerate(dets):
        da[i, :len(d)] = mot_amd.boxes_array(d)
    return fd, torch.from_numpy(da.view(np.uint8).reshape(len(dets), -1)).cuda()


def spin(us):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e6 < us:
        pass


oracle = orc.load_oracle()
nframes = a.frames
scene = synth.Scene(a.n, 80, stream_id=7, miss_pct=a.miss, fp_pct=a.fp)
items = list(scene.frames(nframes))
frames = [f for f, _ in items]; dets = [d[:a.cap] for _, d in items]
fd, dd = dev(frames, dets)
m = orc.OracleMot(oracle, 0, 0, a.cap)
refs = [m.step(frames[f], dets[f]) for f in range(nframes)]
m.close()

ham = None
if a.hammer:
    hs = synth.Scene(1024, 80, stream_id=3)
    hitems = list(hs.frames(6))
    hfd, hdd = dev([f for f, _ in hitems], [d[:1024] for _, d in hitems])
    hn = [min(len(d), 1024) for _, d in hitems]
    ham = mot_amd.MotContext(max_tracks=1024, max_dets=1024)
    hk = 0

def state_of(c, n_live):
    """per live track: (name, bytes) of every piece of device state"""
    out = []
    for i in range(0, n_live, a.state_stride):
        xm, al, pos, sc, first, pend = c.live_model(i)
        rec = dict(pos=pos.tobytes(), scale=sc.tobytes(), flags=bytes([first & 255, pend & 255, (pend >> 8) & 255, 0]))
        if not first:                                                   # a track spawned in this frame has never been predicted: its slot still holds the previous owner's model and response
            rec.update(xm=xm.tobytes(), alpha=al.tobytes(), resp=c.live_response(i).tobytes())
        out.append(rec)
    return out


snap_bytes = 0; snap_buf = None


def snap_report(snaps, cap):
    """frame by frame: where do the recorded live lists / predicted boxes leave the oracle's?"""
    rep_lines = []
    prev_tids = []
    for f in range(nframes):
        w = np.frombuffer(snaps[f].tobytes(), np.int32)
        nl = int(w[0]); o = 4
        tid = w[o:o + cap].view(np.uint32); o += cap
        live = w[o:o + 6 * cap].reshape(cap, 6); o += 6 * cap
        pred = w[o:o + 6 * cap].reshape(cap, 6); o += 6 * cap
        slot = w[o:o + cap]; o += cap
        pos = w[o:o + 6 * cap].reshape(cap, 6); o += 6 * cap
        pend = w[o:o + cap]; o += cap
        first = w[o:o + cap]; o += cap
        upd_slots = w[o:o + 64]; o += 64
        upd_det = w[o:o + 64]; o += 64
        hdr = w[o:o + 64]
        ref = refs[f]
        rl = np.stack([ref["live"][k] for k in ("l", "t", "b", "r")], 1); rp = np.stack([ref["predicted"][k] for k in ("l", "t", "b", "r")], 1) if len(ref["predicted"]) else np.zeros((0, 4), np.int32)
        ok_t = nl == len(ref["tids"]) and np.array_equal(tid[:nl], ref["tids"])
        dl = [i for i in range(min(nl, len(rl))) if not np.array_equal(live[i, :4], rl[i])]
        dp = [i for i in range(min(len(prev_tids), len(rp))) if not np.array_equal(pred[i, :4], rp[i])]
        rep_lines.append(dict(frame=f, nlive=nl, upd_count=int(w[1]), tids_equal=bool(ok_t), live_diff=[(i, int(tid[i]), live[i, :4].tolist(), rl[i].tolist()) for i in dl[:4]],
                              pred_diff=[(i, int(prev_tids[i]), pred[i, :4].tolist(), rp[i].tolist()) for i in dp[:4]],
                              assigned_ref=[int(ref["assigned"][i]) for i in dp[:4]], lap_hdr=hdr[:16].tolist() + hdr[16:32].tolist(),
                              upd_slots=upd_slots[:max(0, min(int(w[1]), 8))].tolist()))
        prev_tids = tid[:nl].tolist()
    return rep_lines


good_trace = None


def trace_diff(bad_t, good_t):
    """records of the wrong run that differ from the last good run's: (kind, frame_no, slot, wrong record, good record)"""
    out = []
    for kind in (0, 1):
        for fr in range(16):
            b, g = bad_t[kind, fr], good_t[kind, fr]
            rows = np.nonzero((b != g).any(axis=1))[0]
            for sl in rows[:6]:
                rb, rg = b[sl].tolist(), g[sl].tolist()
                if kind == 0:
                    f = lambda r: dict(frame_no=r[0], pos=(r[1], r[2]), argmax=r[3] & 0xFFFF, item=r[3] >> 16, peak=float(np.array(r[4], np.int32).view(np.float32)), pend=r[5], first=r[6], newpos=(r[7] & 0xFFFF, r[7] >> 16))
                else:
                    f = lambda r: dict(frame_no=r[0], box=(r[1], r[2]), det_index=r[3], first=r[4], slot=r[5], item=r[6])
                out.append(("predict" if kind == 0 else "update", int(sl), f(rb), f(rg)))
    out.sort(key=lambda x: x[2]["frame_no"])
    return out


bad = []
state0 = {}
t_start = time.time()
for rep in range(a.reps):
    if rep % 100 == 0:
        print(f"rep {rep} ({time.time() - t_start:.0f} s)", file=sys.stderr, flush=True)   # (a crash leaves no result line: where it happened)
    if a.dirty:
        # recycle device memory with garbage in it: the context's allocations below come from pages this process has used before
        g = torch.empty(96 << 20, dtype=torch.int32, device="cuda"); g.random_(-2**31, 2**31 - 1); torch.cuda.synchronize(); del g; torch.cuda.empty_cache()
    c = mot_amd.MotContext(max_tracks=a.cap, max_dets=a.cap)
    if a.snap and snap_buf is None:
        snap_bytes = c.debug_snapshot_bytes()
        snap_buf = torch.zeros((nframes, snap_bytes), dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    lives = []
    for f in range(nframes):
        nxt = f + 1 if f + 1 < nframes else None
        if ham is not None:
            for _ in range(rng.randint(0, 2)):
                ham.step_frame_device(hfd[hk % 6].data_ptr(), hdd[hk % 6].data_ptr(), hn[hk % 6]); hk += 1
        spin(rng.uniform(0, 200))
        if f == 3:
            c.step_frame_device(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]))
        elif f == a.wrong_at and nxt is not None:
            c.step_frame_device_ahead(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]), fd[0].data_ptr(), dd[0].data_ptr(), len(dets[0]))
        else:
            c.step_frame_device_ahead(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]), fd[nxt].data_ptr() if nxt is not None else 0,
                                      dd[nxt].data_ptr() if nxt is not None else 0, len(dets[nxt]) if nxt is not None else 0)
        if a.snap:
            c.debug_snapshot(snap_buf[f].data_ptr())
        if not a.sparse_checks or f == nframes - 1:
            boxes, tids, _ = c.live_tracks()
            ref = refs[f]
            ok = np.array_equal(tids, ref["tids"]) and all(np.array_equal(boxes[k], ref["live"][k]) for k in KEYS)
            if ok and a.state:
                st = state_of(c, len(tids))
                n_state_checks = globals().get("n_state_checks", 0) + 1
                if rep == 0:
                    state0[f] = st
                else:
                    diffs = [(i * a.state_stride, k) for i, (x, y) in enumerate(zip(st, state0[f])) for k in x if x.get(k) != y.get(k)]
                    if diffs or len(st) != len(state0[f]):
                        def mag(i, k):
                            dt = np.float32
                            u, v = np.frombuffer(st[i // a.state_stride][k], dt), np.frombuffer(state0[f][i // a.state_stride][k], dt)
                            return float(np.max(np.abs(u - v))) if (k in ("xm", "alpha", "resp", "scale") and len(u) == len(v)) else -1.0
                        rec = dict(rep=rep, frame=f, kind="state differs from repetition 0", n_diff=len(diffs), first=[(i, k, mag(i, k)) for i, k in diffs[:12]],
                                   lap=c.lap_stats()[:48].tolist(), assoc=c.assoc_stats()[:8].tolist())
                        bad.append(rec)
                        break
            if a.trace and f == nframes - 1:
                tr_now = c.debug_trace()
                if ok:
                    good_trace = tr_now
            if not ok:
                nmin = min(len(boxes), len(ref["live"]))
                idx = [i for i in range(nmin) if any(boxes[k][i] != ref["live"][k][i] for k in KEYS)]
                rec = dict(rep=rep, frame=f, n_diff=len(idx), first=[(i, [int(boxes[k][i]) for k in KEYS], [int(ref["live"][k][i]) for k in KEYS]) for i in idx[:4]],
                           lap=c.lap_stats()[:48].tolist(), assoc=c.assoc_stats()[:8].tolist())
                if a.trace and good_trace is not None:
                    rec["trace_diff"] = trace_diff(tr_now, good_trace)[:12]
                if a.snap:
                    torch.cuda.synchronize()
                    sn = snap_buf.cpu().numpy()
                    rec["by_frame"] = [r for r in snap_report(sn, a.cap) if (r["live_diff"] or r["pred_diff"] or not r["tids_equal"])]
                bad.append(rec)
                os.makedirs(a.dump, exist_ok=True)
                resp = [c.live_response(i) for i in idx[:8]] if idx else []
                np.savez(os.path.join(a.dump, f"mismatch_n{a.n}_rep{rep}_f{f}_{os.getpid()}.npz"), argv=np.array(sys.argv[1:], dtype=object).astype(str),
                         env=np.array([f"{k}={v}" for k, v in os.environ.items() if k.startswith("MOT_")]), rep=rep, frame=f,
                         got=np.stack([boxes[k] for k in KEYS], 1), exp=np.stack([ref["live"][k] for k in KEYS], 1), got_tids=tids, exp_tids=ref["tids"],
                         idx=np.array(idx), resp=np.array(resp), lap=c.lap_stats(), assoc=c.assoc_stats(), snaps=(snap_buf.cpu().numpy() if a.snap else np.zeros(0)))
                break
    c.close()
if ham is not None:
    ham.close()
print(json.dumps(dict(n=a.n, miss=a.miss, fp=a.fp, reps=a.reps, hammer=a.hammer, sparse_checks=a.sparse_checks, dirty=a.dirty,
                      env={k: v for k, v in os.environ.items() if k.startswith("MOT_")}, mismatches=len(bad), detail=bad[:3], seconds=round(time.time() - t_start, 1))))
sys.exit(1 if bad else 0)
