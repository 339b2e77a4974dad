#!/usr/bin/env python3
"""numeric difference of two tools/state_dump.py --npz records: per frame the tracks whose model / alpha / response / pos differ, with magnitudes"""
import sys, re, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
keys = sorted(set(a.files) | set(b.files), key=lambda k: (int(re.match(r"f(\d+)_t(\d+)_", k).group(1)), int(re.match(r"f(\d+)_t(\d+)_", k).group(2)), k))
n = 0
for k in keys:
    if k not in a.files or k not in b.files:
        print("only in one:", k); n += 1; continue
    x, y = a[k], b[k]
    if x.shape != y.shape or not np.array_equal(x, y):
        d = np.abs(x.astype(np.float64) - y.astype(np.float64)) if x.shape == y.shape else None
        rel = float(d.max() / (np.abs(x).max() + 1e-30)) if d is not None else -1
        print(f"{k:28s} differs: max abs {float(d.max()) if d is not None else -1:.3e}  rel-to-max {rel:.3e}  #elements {int((d > 0).sum()) if d is not None else -1} of {x.size}", (x.tolist(), y.tolist()) if x.size <= 4 else "")
        if d is not None and x.size > 400 and n < 3:                      # the pattern of the first few model differences: which planes / bins
            idx = np.nonzero(d > 0)[0] // 2                                 # complex element index = plane * bins + bin
            bins = 220 if x.size == 31 * 220 * 2 else x.size // 62
            runs = []; s0 = prev = int(idx[0])
            for v in idx[1:]:
                v = int(v)
                if v > prev + 1: runs.append((s0, prev)); s0 = v
                prev = v
            runs.append((s0, prev))
            print("      complex-element runs (start, end) [plane.bin]:", [(f"{a}={a // bins}.{a % bins}", f"{b}={b // bins}.{b % bins}") for a, b in runs[:24]], "..." if len(runs) > 24 else "")
        n += 1
        if n > 60:
            print("..."); break
print("records that differ:", n)
