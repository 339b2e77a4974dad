#!/usr/bin/env python3
"""Regression fixtures for the association tiers: every problem on which the device path ever disagreed with the checker.

    python tests/golden/make_assoc_regressions.py [more.npz ...]

Inputs: box lists (trk, det) saved by tools/assign_stress.py when a mismatch occurs (it appends them to
gpurun_out/assign_stress_fail_*.npz).  Output: tests/golden/assoc_regressions.npz holding, per case k, `trk_k`, `det_k` (bbox_t
arrays) and the assignment / cost the REFERENCE's own assignmentoptimal (oracle/_ref/libref_hungarian.so, compiled from
trackers/hungarian/hungarian.cpp) returns for the td.cpp:386-457 cost matrix of those boxes -- so it runs in the build
container only.  Existing cases are kept; new files are appended (boxes only: 45 KB per 1024-line case).

Case 0: round 2, 928 x 928, tier 1 refuses (tie, 18 cyclic nodes), tier 2 accepts; a device-side bug of the sparse emulation's
batched event loop returned a different (equal-cost) assignment (fixed in round 2).
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402

OUT = os.path.join(HERE, "assoc_regressions.npz")


def expected(hung, lib, trk, det):
    nT, nD = len(trk), len(det)
    d = orc.cost_matrix(lib, trk, det)
    nr, nc = (nT, nD) if nT < nD else (nD, nT)                          # rows = the smaller side (td.cpp:462-469)
    a = np.zeros(nr, np.int32); c = C.c_double(0)
    hung.refhung_assign(orc.P(a), C.byref(c), orc.P(d.copy()), nr, nc)
    ao, co = orc.assignment_optimal(lib, d, nr, nc)
    assert np.array_equal(a, ao) and c.value == co, "oracle restatement and reference disagree on a regression case"
    return a, c.value


def main():
    hung = orc.load_ref("hungarian"); lib = orc.load_oracle()
    cases = []
    if os.path.exists(OUT):
        g = np.load(OUT)
        cases = [(g[f"trk_{k}"], g[f"det_{k}"]) for k in range(int(g["n"]))]
    for path in sys.argv[1:]:
        g = np.load(path)
        key = (g["trk"].tobytes(), g["det"].tobytes())
        if all((t.tobytes(), d.tobytes()) != key for t, d in cases):
            cases.append((g["trk"], g["det"]))
    out = {"n": np.int32(len(cases))}
    for k, (trk, det) in enumerate(cases):
        a, c = expected(hung, lib, trk, det)
        out[f"trk_{k}"] = trk; out[f"det_{k}"] = det; out[f"a_{k}"] = a; out[f"c_{k}"] = np.float64(c)
    np.savez_compressed(OUT, **out)
    print(f"{OUT}: {len(cases)} cases")


if __name__ == "__main__":
    main()
