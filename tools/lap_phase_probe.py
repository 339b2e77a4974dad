#!/usr/bin/env python3
"""Where the solver's workgroup (lap_solve_kernel) spends its time on the bench stream: per frame the debug ticks the kernels leave in the LAP header
(10 ns units; MOT_LAP_DEBUG=1 makes mot_get_lap_stats print them).  One synchronisation per frame.  usage: lap_phase_probe.py N FRAMES"""
import os, sys
os.environ["MOT_LAP_DEBUG"] = "1"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mot_amd
from multiple_object_tracking_amd import synth
n, nf = int(sys.argv[1]), int(sys.argv[2])
scene = synth.Scene(n, 80, stream_id=0)
items = list(scene.frames(nf))
frames = torch.from_numpy(np.stack([f for f, _ in items])).cuda()
dets = [d[:1024] for _, d in items]
da = np.zeros((nf, max(len(d) for d in dets)), mot_amd.BBOX_DTYPE)
for i, d in enumerate(dets):
    da[i, :len(d)] = mot_amd.boxes_array(d)
dd = torch.from_numpy(da.view(np.uint8).reshape(nf, -1)).cuda()
c = mot_amd.MotContext(max_tracks=1024, max_dets=1024)
for f in range(nf):
    c.step_frame_device(frames[f].data_ptr(), dd[f].data_ptr(), len(dets[f])); c.sync()
    sys.stderr.write(f"frame {f}: ")
    l = c.lap_stats()
    sys.stderr.write(f"   used={int(l[15])} rounds={int(l[1])} free={int(l[2])} searches={int(l[3])} edges={int(l[5])} solver_ticks={int(l[7])}\n")
