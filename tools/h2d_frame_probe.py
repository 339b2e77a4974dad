#!/usr/bin/env python3
"""per-frame GPU time (events on the context stream) of the same frames in resident and host-fed mode"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mot_amd
from bench import gen_stream
n, nf = 1024, 80
counts = []
fh, dh = gen_stream(n, 80, nf, counts=counts)
pf = torch.from_numpy(fh).pin_memory(); pd = torch.from_numpy(dh.view(np.uint8).reshape(nf, -1)).pin_memory()
fd = torch.from_numpy(fh).cuda(); dd = torch.from_numpy(dh.view(np.uint8).reshape(nf, -1)).cuda()
st = torch.cuda.Stream()
for mode in ("device", "host"):
    c = mot_amd.MotContext(max_tracks=n, max_dets=n, stream=st.cuda_stream)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(nf + 1)]
    with torch.cuda.stream(st):
        for f in range(nf):
            evs[f].record(st)
            if mode == "device": c.step_frame_device(fd[f].data_ptr(), dd[f].data_ptr(), counts[f])
            else: c.step_frame_host(pf[f].data_ptr(), pd[f].data_ptr(), counts[f])
        evs[nf].record(st)
        st.synchronize()
    per = [evs[f].elapsed_time(evs[f + 1]) * 1e3 for f in range(nf)]
    print(mode, "frames 40..79 us:", [round(x) for x in per[40:]], "mean", round(sum(per[40:]) / 40))
    c.close()
