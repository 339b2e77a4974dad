#!/usr/bin/env python3
"""times mot_step_frame_host (frame + detections uploaded inside the loop) against mot_step_frame_device (resident) on the bench stream"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mot_amd
from bench import gen_stream
n, nf = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 90
counts = []
fh, dh = gen_stream(n, 80, nf, counts=counts)
pf = torch.from_numpy(fh).pin_memory(); pd = torch.from_numpy(dh.view(np.uint8).reshape(nf, -1)).pin_memory()
fd = torch.from_numpy(fh).cuda(); dd = torch.from_numpy(dh.view(np.uint8).reshape(nf, -1)).cuda()
st = torch.cuda.Stream()
for mode in ("device", "host"):
    c = mot_amd.MotContext(max_tracks=n, max_dets=n, stream=st.cuda_stream)
    with torch.cuda.stream(st):
        def step(f):
            if mode == "device": c.step_frame_device(fd[f].data_ptr(), dd[f].data_ptr(), counts[f])
            else: c.step_frame_host(pf[f].data_ptr(), pd[f].data_ptr(), counts[f])
        for f in range(30): step(f)
        st.synchronize(); t0 = time.perf_counter()
        for f in range(30, nf): step(f)
        st.synchronize(); dt = time.perf_counter() - t0
    print(f"{mode}: {dt / (nf - 30) * 1e6:.1f} us/frame, MOT_H2D_MODE={os.environ.get('MOT_H2D_MODE', '0')}")
    c.close()
# host-side duration of the calls (does the upload block the host?)
c = mot_amd.MotContext(max_tracks=n, max_dets=n, stream=st.cuda_stream)
with torch.cuda.stream(st):
    for f in range(30): c.step_frame_host(pf[f].data_ptr(), pd[f].data_ptr(), counts[f])
    st.synchronize()
    ts = []
    for f in range(30, 60):
        t0 = time.perf_counter(); c.step_frame_host(pf[f].data_ptr(), pd[f].data_ptr(), counts[f]); ts.append((time.perf_counter() - t0) * 1e6)
    st.synchronize()
print("host-side call durations (us):", [round(x) for x in ts[:12]])
hp = torch.empty(2764800, dtype=torch.uint8).pin_memory(); dv = torch.empty(2764800, dtype=torch.uint8, device="cuda")
import ctypes as C
hip = C.CDLL("libamdhip64.so")
attr = (C.c_int * 64)()
print("hipPointerGetAttributes rc", hip.hipPointerGetAttributes(C.byref(attr), C.c_void_p(pf[5].data_ptr())), "type", attr[0])
