#!/usr/bin/env python3
"""Prints one sha256 per frame over the complete device state of every live track of the device-resident loop (model, alpha, pos, scale, flags,
response map; boxes, ids) for a seeded noisy stream -- two builds / kernel variants / runs are equal iff their outputs are (GPU box).
usage: state_dump.py N CAP MISS FP FRAMES [STREAM_ID] [--ahead] [--final-only] [--size S]   (--final-only: nothing is read back -- or synchronised -- before the last frame)"""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mot_amd
from multiple_object_tracking_amd import synth

n, cap, miss, fp, nframes = (int(x) for x in sys.argv[1:6])
sid = int(sys.argv[6]) if len(sys.argv) > 6 and not sys.argv[6].startswith("-") else 21
ahead = "--ahead" in sys.argv
final_only = "--final-only" in sys.argv
npz = sys.argv[sys.argv.index("--npz") + 1] if "--npz" in sys.argv else None
size = int(sys.argv[sys.argv.index("--size") + 1]) if "--size" in sys.argv else 80
full = {}
scene = synth.Scene(n, size, stream_id=sid, miss_pct=miss, fp_pct=fp)
items = list(scene.frames(nframes))
frames = [f for f, _ in items]; dets = [d[:cap] for _, d in items]
fd = torch.from_numpy(np.stack(frames)).cuda()
nmax = max(len(d) for d in dets)
da = np.zeros((nframes, max(nmax, 1)), mot_amd.BBOX_DTYPE)
for i, d in enumerate(dets):
    da[i, :len(d)] = mot_amd.boxes_array(d)
dd = torch.from_numpy(da.view(np.uint8).reshape(nframes, -1)).cuda()
c = mot_amd.MotContext(max_tracks=cap, max_dets=cap, dev_size=size) if size != 80 else mot_amd.MotContext(max_tracks=cap, max_dets=cap)
for f in range(nframes):
    if ahead and f + 1 < nframes:
        c.step_frame_device_ahead(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]), fd[f + 1].data_ptr(), dd[f + 1].data_ptr(), len(dets[f + 1]))
    else:
        c.step_frame_device(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]))
    if final_only and f + 1 < nframes:
        continue
    boxes, tids, ages = c.live_tracks()
    h = hashlib.sha256()
    h.update(boxes.tobytes()); h.update(tids.tobytes()); h.update(ages.tobytes())
    for i in range(len(tids)):
        xm, al, pos, sc, first, pend = c.live_model(i)
        resp = c.live_response(i)
        if first:                                                       # never predicted: the slot still holds its previous owner's model / response
            xm = np.zeros_like(xm); al = np.zeros_like(al); resp = np.zeros_like(resp)
        for part in (xm, al, pos, sc, resp):
            h.update(np.ascontiguousarray(part).tobytes())
        h.update(bytes([first & 255, pend & 255, (pend >> 8) & 255]))
        if npz:
            full[f"f{f}_t{int(tids[i])}_xm"] = xm.view(np.float32).copy(); full[f"f{f}_t{int(tids[i])}_alpha"] = al.copy(); full[f"f{f}_t{int(tids[i])}_resp"] = resp.copy()
            full[f"f{f}_t{int(tids[i])}_pos"] = np.array([pos[k] for k in ("l", "t", "b", "r")]); full[f"f{f}_t{int(tids[i])}_flags"] = np.array([first, pend])
    upd = int(c.assoc_stats()[0]) if False else 0
    print(f"frame {f} live {len(tids)} {h.hexdigest()}")
l = c.lap_stats()
print(f"stats certified {int(l[16])} tie {int(l[20])} provisional {int(l[21])} swaps {int(l[22])} dense_bits {int(l[23])} sparse_accepted {int(l[24])} sparse_refused {int(l[25])}")
c.close()
if npz:
    np.savez_compressed(npz, **full)
