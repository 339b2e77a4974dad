#!/usr/bin/env python3
"""The comparison north_star asks for: the 31-channel r2c transforms of one launch done UNFUSED by rocFFT (through
torch.fft.rfft2 = hipFFT/rocFFT on ROCm) on feature planes that already sit in HBM, against the fused in-kernel transforms.
usage: rocfft_compare.py   (GPU box)"""
import torch

def bench(batch, h, w, reps=50):
    x = torch.randn(batch, h, w, device="cuda", dtype=torch.float32)
    y = torch.fft.rfft2(x); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y = torch.fft.rfft2(x)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    mb = (x.numel() * 4 + y.numel() * 8) / 1e6
    return us, mb

for name, tracks, cells in (("1024 tracks, 80x80 templates (20x20 cells)", 1024, 20), ("256 tracks, 148x148 templates (37x37 cells)", 256, 37),
                            ("256 tracks, 164x164 templates (41x41 cells)", 256, 41)):
    us, mb = bench(tracks * 31, cells, cells)
    print(f"{name}: rocFFT rfft2 of {tracks * 31} planes {cells}x{cells}: {us:.1f} us per launch, {mb:.1f} MB of HBM traffic (planes in, half spectra out) "
          f"= {mb / us:.2f} TB/s")
