#!/usr/bin/env python3
"""Long run of the device-resident loop against the CPU oracle's frame loop (GPU box; the oracle is the slow side).
usage: devloop_soak.py N_TRACKS FRAMES [MISS_PCT FP_PCT]   -- every frame: track ids and live boxes must be equal"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import mot_amd, orc
from bench import gen_stream

n, nf = int(sys.argv[1]), int(sys.argv[2])
miss, fp = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 0)
lib = orc.load_oracle()
counts = []
frames, dets = gen_stream(n, 80, nf, miss_pct=miss, fp_pct=fp, nms=bool(miss or fp), counts=counts)
if not counts: counts = [n] * nf
fd = torch.from_numpy(frames).cuda(); dd = torch.from_numpy(dets.view(np.uint8).reshape(nf, -1)).cuda()
c = mot_amd.MotContext(max_tracks=1024, max_dets=1024)
m = orc.OracleMot(lib, 0, 0, 1024)
used = [0, 0, 0]; t0 = time.time()
for f in range(nf):
    c.step_frame_device(fd[f].data_ptr(), dd[f].data_ptr(), counts[f])
    ref = m.step(frames[f], dets[f][:counts[f]])
    boxes, tids, _ = c.live_tracks()
    if f > 0: used[int(c.lap_stats()[15])] += 1
    bnp = lambda b: np.stack([b[k] for k in ("l", "t", "b", "r", "type")], axis=1)
    ok = np.array_equal(tids, ref["tids"]) and np.array_equal(bnp(boxes), bnp(ref["live"]))
    if not ok:
        print(f"MISMATCH at frame {f}"); sys.exit(1)
    if f % 10 == 0: print(f"frame {f} ok ({time.time() - t0:.0f} s), decided by certificate / sparse / dense = {used}", flush=True)
print(f"devloop_soak OK: {nf} frames of {n} tracks, decided by certificate / sparse emulation / dense emulation = {used}")
