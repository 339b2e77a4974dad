#!/usr/bin/env python3
"""short resident run with look-ahead for a rocprofv3 --kernel-trace timeline (steady-state frames)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mot_amd
from bench import gen_stream
n, nf = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 70
counts = []
fh, dh = gen_stream(n, 80, nf, counts=counts)
fd = torch.from_numpy(fh).cuda(); dd = torch.from_numpy(dh.view(np.uint8).reshape(nf, -1)).cuda()
st = torch.cuda.Stream()
c = mot_amd.MotContext(max_tracks=n, max_dets=n, stream=st.cuda_stream)
with torch.cuda.stream(st):
    for f in range(nf - 1):
        c.step_frame_device_ahead(fd[f].data_ptr(), dd[f].data_ptr(), counts[f], fd[f + 1].data_ptr(), dd[f + 1].data_ptr(), counts[f + 1])
    st.synchronize()
c.close()
