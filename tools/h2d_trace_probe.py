#!/usr/bin/env python3
"""short host-fed run for a rocprofv3 --kernel-trace --memory-copy-trace timeline"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mot_amd
from bench import gen_stream
n, nf = 1024, 40
counts = []
fh, dh = gen_stream(n, 80, nf, counts=counts)
pf = torch.from_numpy(fh).pin_memory(); pd = torch.from_numpy(dh.view(np.uint8).reshape(nf, -1)).pin_memory()
st = torch.cuda.Stream()
c = mot_amd.MotContext(max_tracks=n, max_dets=n, stream=st.cuda_stream)
with torch.cuda.stream(st):
    for f in range(nf): c.step_frame_host(pf[f].data_ptr(), pd[f].data_ptr(), counts[f])
    st.synchronize()
c.close()
