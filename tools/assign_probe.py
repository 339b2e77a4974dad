#!/usr/bin/env python3
"""Association stage alone (no KCF kernels beside it): mot_assign on tracking-like box sets; for rocprofv3 --kernel-trace --stats.
usage: assign_probe.py N REPS [SPREAD_PX]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mot_amd

n, reps = int(sys.argv[1]), int(sys.argv[2])
spread = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rng = np.random.default_rng(3)
c = mot_amd.MotContext(max_tracks=1024, max_dets=1024)
for it in range(reps):
    cx = rng.integers(40, 1240, n); cy = rng.integers(40, 680, n)
    trk = [(int(cx[i] + rng.integers(-spread, spread + 1)) - 40, int(cy[i] + rng.integers(-spread, spread + 1)) - 40, int(cy[i]) + 39, int(cx[i]) + 39, i % 3, 0.9) for i in range(n)]
    det = [(int(cx[i] + rng.integers(-2, 3)) - 40, int(cy[i] + rng.integers(-2, 3)) - 40, int(cy[i]) + 39, int(cx[i]) + 39, int(i % 3), 0.9) for i in rng.permutation(n)]
    c.assign(trk, det)
    print(it, c.lap_stats()[:16].tolist())
