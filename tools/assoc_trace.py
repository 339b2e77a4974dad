#!/usr/bin/env python3
"""Thread 0's time line through the sparse emulation's cycles of one frame (GPU box, MOT_MK_TIMING=1).
usage: assoc_trace.py N FRAME [CYCLES]
tags: 1 cycle top, 2 after the phase-start work, 11 / 19 end of a one-event / batch iteration, 3 end of the iteration that found nothing,
4 augmentation done, 5 phase end, 6 behind barrier A, 7 wave minimum ready, 8 behind barrier B, 9 update done"""
import os, sys
os.environ["MOT_MK_TIMING"] = "1"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mot_amd
from multiple_object_tracking_amd import synth

n, fr = int(sys.argv[1]), int(sys.argv[2]); ncyc = int(sys.argv[3]) if len(sys.argv) > 3 else 12
scene = synth.Scene(n, 80, stream_id=0)
items = list(scene.frames(fr + 1))
frames = torch.from_numpy(np.stack([f for f, _ in items])).cuda()
da = np.zeros((fr + 1, 1024), mot_amd.BBOX_DTYPE)
for i, (_, d) in enumerate(items):
    da[i, :len(d)] = mot_amd.boxes_array(d)
dd = torch.from_numpy(da.view(np.uint8).reshape(fr + 1, -1)).cuda()
c = mot_amd.MotContext(max_tracks=1024, max_dets=1024)
for f in range(fr + 1):
    c.profile_frame_device(frames[f].data_ptr(), dd[f].data_ptr(), len(items[f][1]))
t = c.assoc_trace()
print("lap", c.lap_stats()[:16].tolist(), "trace entries", len(t))
tags, ts = t[:, 0], t[:, 1]
tops = np.nonzero(tags == 1)[0]
names = {1: "top", 2: "start", 3: "none", 11: "ev", 19: "batch", 4: "aug", 5: "end", 6: "A", 7: "min", 8: "B", 9: "upd", 20: "path", 21: "migr", 22: "reset", 23: "rt1", 24: "rt2", 25: "srch"}
for ci in range(min(ncyc, len(tops) - 1)):
    lo, hi = tops[ci + 40 if len(tops) > ncyc + 41 else ci], None
    k = list(tops).index(lo)
    hi = tops[k + 1]
    seg = " ".join(f"{names.get(int(tags[i]), int(tags[i]))}+{(ts[i] - ts[lo]) * 10}" for i in range(lo + 1, hi + 1))
    print(f"cycle {k}: {seg}  (ns)")
# averages per segment kind over all cycles
import collections
acc = collections.defaultdict(list)
for k in range(len(tops) - 1):
    prev = tops[k]
    for i in range(tops[k] + 1, tops[k + 1] + 1):
        acc[(names.get(int(tags[prev]), int(tags[prev])), names.get(int(tags[i]), int(tags[i])))].append((ts[i] - ts[prev]) * 10)
        prev = i
for key, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{key[0]:>6} -> {key[1]:<6} n={len(v):5d} mean {np.mean(v):7.0f} ns  total {sum(v) / 1000:8.1f} us")
