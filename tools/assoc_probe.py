#!/usr/bin/env python3
"""Per-frame association statistics of the device-resident loop on a synthetic stream (GPU box).
usage: assoc_probe.py N FRAMES [MISS_PCT FP_PCT]"""
import os, sys
os.environ.setdefault("MOT_MK_TIMING", "1")   # per-cycle clock reads inside the emulation (off by default: they cost a few % of its time)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mot_amd
from multiple_object_tracking_amd import synth

n, nf = int(sys.argv[1]), int(sys.argv[2])
miss, fp = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 0)
scene = synth.Scene(n, 80, stream_id=0 if not miss else 5, miss_pct=miss, fp_pct=fp)
items = list(scene.frames(nf))
frames = torch.from_numpy(np.stack([f for f, _ in items])).cuda()
dets = [d[:1024] for _, d in items]
nmax = max(len(d) for d in dets)
da = np.zeros((nf, nmax), mot_amd.BBOX_DTYPE)
for i, d in enumerate(dets):
    da[i, :len(d)] = mot_amd.boxes_array(d)
dd = torch.from_numpy(da.view(np.uint8).reshape(nf, -1)).cuda()
c = mot_amd.MotContext(max_tracks=1024, max_dets=1024)
for f in range(nf):
    ms = c.profile_frame_device(frames[f].data_ptr(), dd[f].data_ptr(), len(dets[f]))
    a = c.assoc_stats(); l = c.lap_stats()
    print(f"f{f} nD={len(dets[f])} live={c.live_count()} stage_ms={np.round(ms, 3).tolist()} lap[outcome,rounds,free,searches,commits,edges,cyc,ticks]={l[:8].tolist()} sparse[status,aug,s5,events,us_s3,us_s5,us_total,used]={l[8:12].tolist() + [int(l[12]) // 100, int(l[13]) // 100, int(l[14]) // 100, int(l[15])]} "
          f"dense_solver[steps,free,us,ran_total,certified_total]={[int(l[26]), int(l[27]), int(l[28]) // 100, int(l[29]), int(l[30])]} "
          f"munkres[s4,s5,sweeps]={a[:3].tolist()} mk_total_us={a[12] / 100:.0f} init_us={a[8] / 100:.0f} s3_us={a[9] / 100:.0f} s5_us={a[11] / 100:.0f}")
print("cumulative outcomes", c.lap_stats()[16:28].tolist())
