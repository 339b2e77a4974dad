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
