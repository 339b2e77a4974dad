#!/usr/bin/env python3
"""Repeats the look-ahead / wrong-announcement scenario of tests/test_gpu_devloop.py::test_device_loop_lookahead_vs_oracle and counts
frames whose live list differs from the oracle (GPU box).  usage: lookahead_flake_probe.py N MISS FP REPS"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import mot_amd, orc
from multiple_object_tracking_amd import synth

n, miss, fp, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
wrong_at = int(os.environ.get("WRONG_AT", "5"))
oracle = orc.load_oracle()
nframes = 9
scene = synth.Scene(n, 80, stream_id=7, miss_pct=miss, fp_pct=fp)
items = list(scene.frames(nframes))
frames = [f for f, _ in items]; dets = [d[:1024] for _, d in items]
fd = torch.from_numpy(np.stack(frames)).cuda()
nmax = max(len(d) for d in dets)
da = np.zeros((len(dets), max(nmax, 1)), mot_amd.BBOX_DTYPE)
for i, d in enumerate(dets):
    da[i, :len(d)] = mot_amd.boxes_array(d)
dd = torch.from_numpy(da.view(np.uint8).reshape(len(dets), -1)).cuda()
m = orc.OracleMot(oracle, 0, 0, 1024)
refs = [m.step(frames[f], dets[f]) for f in range(nframes)]
m.close()
bad = {}
for rep in range(reps):
    c = mot_amd.MotContext(max_tracks=1024, max_dets=1024)
    for f in range(nframes):
        nxt = f + 1 if f + 1 < nframes else None
        if f == 3:
            c.step_frame_device(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]))
        elif f == wrong_at and nxt is not None:
            c.step_frame_device_ahead(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]), fd[0].data_ptr(), dd[0].data_ptr(), len(dets[0]))
        else:
            c.step_frame_device_ahead(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]), fd[nxt].data_ptr() if nxt is not None else 0,
                                      dd[nxt].data_ptr() if nxt is not None else 0, len(dets[nxt]) if nxt is not None else 0)
        if os.environ.get("EVERY_FRAME", "1") == "1" or f == nframes - 1:
            boxes, tids, _ = c.live_tracks()
            ref = refs[f]
            ok = np.array_equal(tids, ref["tids"]) and all(np.array_equal(boxes[k], ref["live"][k]) for k in ("l", "t", "b", "r"))
            if not ok:
                idx = [i for i in range(min(len(boxes), len(ref["live"]))) if any(boxes[k][i] != ref["live"][k][i] for k in ("l", "t", "b", "r"))]
                bad.setdefault(rep, []).append((f, len(idx), [(int(boxes["l"][i]), int(boxes["t"][i]), int(ref["live"]["l"][i]), int(ref["live"]["t"][i])) for i in idx[:3]]))
    c.close()
print(f"n={n} reps={reps} wrong_at={wrong_at}: {len(bad)} runs end with a live list different from the oracle: {bad}")
