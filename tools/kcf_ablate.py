#!/usr/bin/env python3
"""What each phase of the 80 x 80 px predict kernel costs the LAUNCH (GPU box; needs `make -C multiple-object-tracking_amd/csrc ablate`).

The probe build (libmot_amd_ablate.so) can skip phases of the KCF kernels.  For every mask: a fresh context runs WARM ordinary frames of the bench
stream, then ONE frame with the mask set, and the duration of that frame's predict launch is read from the kernel's own begin / end stamps
(mot_debug_predict_timing).  The difference to mask 0 is what the phase costs a launch of 1024 workgroups under real contention (two workgroups
per CU, all phases of the neighbours running) -- unlike workgroup 0's phase stamps, which are one workgroup's latencies.
usage: kcf_ablate.py [--tracks 1024] [--reps 8] [--warm 10]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MOT_AMD_LIB"] = os.path.join(ROOT, "multiple-object-tracking_amd", "libmot_amd_ablate.so")
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mot_amd
from bench import gen_stream

PHASES = ["blend prologue", "crop", "gradient", "histogram", "energy+norm", "channels", "forward FFT", "correlation", "inverse FFT", "response store + arg-max", "model prefetch"]
ap = argparse.ArgumentParser()
ap.add_argument("--tracks", type=int, default=1024); ap.add_argument("--reps", type=int, default=8); ap.add_argument("--warm", type=int, default=10)
ap.add_argument("--size", type=int, default=80); ap.add_argument("--only-base", action="store_true")
a = ap.parse_args()
lib = mot_amd.load_library()
nf = a.warm + 2
frames_h, dets_h = gen_stream(a.tracks, a.size, nf)
fd = torch.from_numpy(frames_h).cuda(); dd = torch.from_numpy(dets_h.view(np.uint8).reshape(nf, -1)).cuda()
fb, db = 720 * 1280 * 3, dd.shape[1]


def sample(mask):
    c = mot_amd.MotContext(max_tracks=a.tracks, max_dets=a.tracks, dev_size=a.size)
    for f in range(a.warm):
        c.step_frame_device_ahead(fd.data_ptr() + f * fb, dd.data_ptr() + f * db, a.tracks, fd.data_ptr() + (f + 1) * fb, dd.data_ptr() + (f + 1) * db, a.tracks)
    c.sync()
    c.debug_predict_timing(1)
    lib.mot_debug_kcf_ablate(int(mask))
    f = a.warm
    c.step_frame_device_ahead(fd.data_ptr() + f * fb, dd.data_ptr() + f * db, a.tracks, fd.data_ptr() + (f + 1) * fb, dd.data_ptr() + (f + 1) * db, a.tracks)
    c.sync()
    lib.mot_debug_kcf_ablate(0)
    t = c.debug_predict_times()
    c.close()
    return float(t[0]) * 1e3


if os.environ.get("MOT_KCF_ONE_PER_CU") == "1":
    print("ONE workgroup per CU (MOT_KCF_ONE_PER_CU=1)")
masks = [0] + [1 << b for b in range(len(PHASES))] + [0b111111111110, 0b01111111110, (1 << 11) - 1]
names = ["(nothing skipped)"] + PHASES + ["everything but the blend + prefetch", "everything but blend", "everything"]
if a.only_base:
    masks, names = masks[:1] + masks[-3:], names[:1] + names[-3:]
base = None
for m, nm in zip(masks, names):
    v = np.array([sample(m) for _ in range(a.reps)])
    med = float(np.median(v))
    if base is None:
        base = med
    print(f"mask {m:#06x}  {nm:38s} predict launch median {med:7.1f} us  (min {v.min():6.1f}, max {v.max():6.1f})   saves {base - med:6.1f} us")
