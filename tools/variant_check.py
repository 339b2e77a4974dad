#!/usr/bin/env python3
"""Runs two noisy streams through the device-resident loop under whatever MOT_* switches the environment holds and compares every frame's live list
(boxes, types, track ids) with the oracle's; exit code 1 on a mismatch.  One process = one combination of switches (they are read once):
tests/test_gpu_variants.py runs a pairwise-covering set of combinations (round-4 verdict item 9).  GPU box only."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import mot_amd, orc
from multiple_object_tracking_amd import synth

KEYS = ("l", "t", "b", "r", "type")
oracle = orc.load_oracle()
bad = 0
# (round 6) the bench stream itself -- tie frames, provisional commits where the switches allow them -- for the combinations that concern them (and the default one)
STREAMS = [(48, 128, 8, 4, 8, 21, False), (300, 1024, 6, 4, 6, 5, True), (140, 1024, 0, 0, 5, 2, True)]
if not any(k.startswith("MOT_") for k in os.environ) or any(k.startswith("MOT_PROV") for k in os.environ):
    STREAMS.append((1024, 1024, 0, 0, 8, 0, True))
for (n, cap, miss, fp, nframes, sid, ahead) in STREAMS:
    scene = synth.Scene(n, 80, stream_id=sid, miss_pct=miss, fp_pct=fp)
    items = list(scene.frames(nframes))
    frames = [f for f, _ in items]; dets = [d[:cap] for _, d in items]
    fd = torch.from_numpy(np.stack(frames)).cuda()
    da = np.zeros((nframes, max(max(len(d) for d in dets), 1)), mot_amd.BBOX_DTYPE)
    for i, d in enumerate(dets):
        da[i, :len(d)] = mot_amd.boxes_array(d)
    dd = torch.from_numpy(da.view(np.uint8).reshape(nframes, -1)).cuda()
    c = mot_amd.MotContext(max_tracks=cap, max_dets=cap)
    m = orc.OracleMot(oracle, 0, 0, cap)
    for f in range(nframes):
        if ahead and f + 1 < nframes:
            c.step_frame_device_ahead(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]), fd[f + 1].data_ptr(), dd[f + 1].data_ptr(), len(dets[f + 1]))
        else:
            c.step_frame_device(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]))
        ref = m.step(frames[f], dets[f])
        boxes, tids, _ = c.live_tracks()
        ok = np.array_equal(tids, ref["tids"]) and all(np.array_equal(boxes[k], ref["live"][k]) for k in KEYS)
        if not ok:
            bad += 1
            print(f"MISMATCH n={n} frame {f}: lap {c.lap_stats()[:16].tolist()}")
            break
    m.close(); c.close()
print("variant_check:", "FAILED" if bad else "ok", {k: v for k, v in os.environ.items() if k.startswith("MOT_")})
sys.exit(1 if bad else 0)
