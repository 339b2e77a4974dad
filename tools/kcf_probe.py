#!/usr/bin/env python3
"""kcf_probe.py -- times the KCF launches of the device-resident loop in isolation (GPU box only).

    python tools/kcf_probe.py [--tracks 1024] [--size 80] [--frames 12]

Prints, per frame, the duration of the predict launch (HIP events around mot_step_begin_device) and of the rest of the
frame, and the phase stamps of workgroup 0 of the predict / update kernels (mot_debug_kcf_phases: crop, gradient,
histogram, energy + norm, half 0, half 1, correlation + inverse + arg-max) in microseconds.  Workgroup 0 shares its CU with
a second workgroup of the same launch, so the stamps are the CONTENDED phase times.
MOT_DBG_EXTRA=1 adds, per frame on stderr, the sub-phase stamps of the R1-resident HBM-slab kernels (148 px) in microseconds after stamp 0: [8] / [9] end of
the first / second half's transforms, [10] / [11] first stripe's gradient / histogram done, [12]..[15] first channel tile (start, channels done, transform done,
barrier), [16] gray conversion of the crop done, [17] correlation done, [18] inverse transform done, [19] arg-max done.
"""
import argparse, importlib, os, sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mot = importlib.import_module("multiple-object-tracking_amd")
from bench import gen_stream  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tracks", type=int, default=1024)
    ap.add_argument("--size", type=int, default=80)
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--det-sizes", type=int, nargs=2, default=None)
    a = ap.parse_args()
    frames_h, dets_h = gen_stream(a.tracks, a.size, a.frames, det_sizes=tuple(a.det_sizes) if a.det_sizes else None)
    fd = torch.from_numpy(frames_h).cuda()
    dd = torch.from_numpy(dets_h.view(np.uint8).reshape(a.frames, -1)).cuda()
    st = torch.cuda.Stream()
    c = mot.MotContext(max_tracks=a.tracks, max_dets=a.tracks, stream=st.cuda_stream, dev_size=a.size)
    c.debug_kcf_phases(True)
    fb, db = 720 * 1280 * 3, dd.shape[1]
    tick = 1e6 / 100e6                                            # wall_clock64: 100 MHz
    pre, rest = [], []
    with torch.cuda.stream(st):
        for f in range(a.frames):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record(st)
            c.step_begin_device(fd.data_ptr() + f * fb)
            e[1].record(st)
            c.step_finish_device(0, dd.data_ptr() + f * db, a.tracks)
            e[2].record(st)
            st.synchronize()
            p8, u8 = c.debug_kcf_phases(True)
            pp = [(p8[i + 1] - p8[i]) * tick for i in range(7)]
            uu = [(u8[i + 1] - u8[i]) * tick for i in range(7)]
            pre.append(e[0].elapsed_time(e[1]) * 1e3); rest.append(e[1].elapsed_time(e[2]) * 1e3)
            print(f"frame {f:3d}  predict launch {pre[-1]:7.1f} us  rest {rest[-1]:8.1f} us | predict wg0 phases " +
                  " ".join(f"{x:5.1f}" for x in pp) + f" = {sum(pp):6.1f} | update wg0 " + " ".join(f"{x:5.1f}" for x in uu))
    print(f"median predict launch {np.median(pre[2:]):.1f} us, rest {np.median(rest[2:]):.1f} us")
    c.close()


if __name__ == "__main__":
    main()
