#!/usr/bin/env python3
"""Host-fed loop (mot_step_frame_host) alone or behind a resident context in the same process, like bench.py's h2d_inclusive window.
usage: hostfed_probe.py N FRAMES WARMUP [resident_first]   -- env MOT_PROV / MOT_EMU_STREAM for A/B runs"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mot_amd
from bench import gen_stream
n, nf, warm = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
res_first = len(sys.argv) > 4 and sys.argv[4] == "1"
counts = []
fh, dh = gen_stream(n, 80, nf + 1, counts=counts)
st = torch.cuda.Stream()
keep = None
if res_first:
    fd = torch.from_numpy(fh).cuda(); dd = torch.from_numpy(dh.view(np.uint8).reshape(nf + 1, -1)).cuda()
    keep = mot_amd.MotContext(max_tracks=n, max_dets=n, stream=st.cuda_stream)
    for f in range(nf):
        keep.step_frame_device_ahead(fd[f].data_ptr(), dd[f].data_ptr(), counts[f], fd[f + 1].data_ptr(), dd[f + 1].data_ptr(), counts[f + 1])
    st.synchronize(); torch.cuda.synchronize()
pf = torch.from_numpy(fh[:nf]).pin_memory(); pd = torch.from_numpy(dh[:nf].view(np.uint8).reshape(nf, -1)).pin_memory()
scratch = torch.empty_like(pf, device="cuda"); scratch.copy_(pf); del scratch
c = mot_amd.MotContext(max_tracks=n, max_dets=n, stream=st.cuda_stream)
for k in range(warm + 1):
    c.step_frame_host(pf[k].data_ptr(), pd[k].data_ptr(), counts[k])
st.synchronize(); torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(warm + 1, nf):
    c.step_frame_host(pf[k].data_ptr(), pd[k].data_ptr(), counts[k])
st.synchronize(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"host-fed MOT_PROV={os.environ.get('MOT_PROV', '1')} MOT_EMU_STREAM={os.environ.get('MOT_EMU_STREAM', '-')} resident_first={int(res_first)}: {dt / (nf - warm - 1) * 1e3:.4f} ms/frame, {n * (nf - warm - 1) / dt / 1e6:.3f} M updates/s")
c.close()
if keep: keep.close()
