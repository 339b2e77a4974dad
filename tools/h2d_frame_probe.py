#!/usr/bin/env python3
"""per-frame GPU time (events on the context stream) of host-fed frames BEHIND a resident run, like bench.py's h2d window"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mot_amd
from bench import gen_stream
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nres, nh = 60, 50
nf = nres + nh
counts = []
fh, dh = gen_stream(n, 80, nf, counts=counts)
fd = torch.from_numpy(fh).cuda(); dd = torch.from_numpy(dh.view(np.uint8).reshape(nf, -1)).cuda()
fb, db = 720 * 1280 * 3, dd.shape[1]
st = torch.cuda.Stream()
c = mot_amd.MotContext(max_tracks=n, max_dets=n, stream=st.cuda_stream)
with torch.cuda.stream(st):
    for f in range(nres):
        c.step_frame_device_ahead(fd[f].data_ptr(), dd[f].data_ptr(), counts[f], fd[f + 1].data_ptr(), dd[f + 1].data_ptr(), counts[f + 1])
    st.synchronize()
    pf = torch.from_numpy(fh[nres:]).pin_memory(); pd = torch.from_numpy(dh[nres:].view(np.uint8).reshape(nh, -1)).pin_memory()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(nh + 1)]
    host = []
    t00 = time.perf_counter()
    for k in range(nh):
        evs[k].record(st)
        t0 = time.perf_counter(); c.step_frame_host(pf[k].data_ptr(), pd[k].data_ptr(), counts[nres + k]); host.append((time.perf_counter() - t0) * 1e6)
    evs[nh].record(st)
    st.synchronize()
    wall = (time.perf_counter() - t00) / nh * 1e6
per = [evs[k].elapsed_time(evs[k + 1]) * 1e3 for k in range(nh)]
print("gpu us/frame:", [round(x) for x in per])
print("host call us:", [round(x) for x in host])
print("wall per frame", round(wall))
c.close()
