#!/usr/bin/env python3
"""Provisional commits on the bench stream (GPU box): N tracks, FRAMES frames through mot_step_frame_device_ahead without any synchronisation in
between, then the cumulative counters (mot_get_lap_stats()[16..28]: certificate outcomes, provisional commits / swaps / dense decisions, sparse
emulation accepted / refused) and the loop's wall time.  Run it with MOT_PROV=0 for the A/B.  usage: prov_probe.py N FRAMES [WARMUP]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mot_amd
from multiple_object_tracking_amd import synth

n, nf = int(sys.argv[1]), int(sys.argv[2])
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 5
scene = synth.Scene(n, 80, stream_id=0)
items = list(scene.frames(nf))
frames = torch.from_numpy(np.stack([f for f, _ in items])).cuda()
dets = [d[:1024] for _, d in items]
nmax = max(len(d) for d in dets)
da = np.zeros((nf, nmax), mot_amd.BBOX_DTYPE)
for i, d in enumerate(dets):
    da[i, :len(d)] = mot_amd.boxes_array(d)
dd = torch.from_numpy(da.view(np.uint8).reshape(nf, -1)).cuda()
c = mot_amd.MotContext(max_tracks=1024, max_dets=1024)
fb, db = frames[0].numel(), dd[0].numel()
def step(f):
    nxt = (frames.data_ptr() + (f + 1) * fb, dd.data_ptr() + (f + 1) * db, len(dets[f + 1])) if f + 1 < nf else (0, 0, 0)
    c.step_frame_device_ahead(frames.data_ptr() + f * fb, dd.data_ptr() + f * db, len(dets[f]), *nxt)
if os.environ.get("PROV_PROBE_PER_FRAME"):
    # one synchronisation per frame: which tier decided, and why a tie frame was (not) committed provisionally (mot_get_lap_stats()[31])
    why = {0: "-", 1: "provisional", 2: "free column in the core", 3: "too many pairs", 4: "not disjoint two-row cycles", 5: "emulation not started", 6: "no core row"}
    prev = np.zeros(32, np.int64)
    for f in range(nf):
        step(f); c.sync()
        l = c.lap_stats().astype(np.int64)
        tie = l[20] - prev[20]
        w = int(l[31])
        reason = "-" if not tie else str(why.get(w & 0xFF, w & 0xFF)) + " (core rows %d, pairs %d)" % ((w >> 8) & 0xFFF, w >> 20)
        print(f"f{f}: used={int(l[15])} tie={int(tie)} edges={int(l[5])} why={reason} aug={int(l[9])} s5={int(l[10])} provisional_total={int(l[21])} swaps={int(l[22])}")
        prev = l
    sys.exit(0)
for f in range(warm + 1):
    step(f)
c.sync(); torch.cuda.synchronize()
t0 = time.perf_counter()
for f in range(warm + 1, nf):
    step(f)
c.sync(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
l = c.lap_stats()
print(f"MOT_PROV={os.environ.get('MOT_PROV', '1')} tracks={n} frames {warm + 1}..{nf - 1}: {dt / (nf - warm - 1) * 1e3:.4f} ms/frame, {n * (nf - warm - 1) / dt / 1e6:.3f} M updates/s")
print(f"  certificate outcomes [certified, gave up, infeasible, too many edges, tie] = {l[16:21].tolist()}; provisional commits / swaps / dense decisions = {l[21:24].tolist()}; "
      f"sparse emulation accepted / refused = {l[24:26].tolist()}")
