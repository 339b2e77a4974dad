#!/usr/bin/env python3
"""Stress of the association tiers against the oracle's order-exact Munkres (GPU box): random crowded scenes with
duplicated centroids (tied optima), rectangular shapes, every size from 97 to 1024 lines.
usage: assign_stress.py SECONDS [SEED] [MISSFP]   -- MISSFP=1: every scene also drops 2-6 % of the detections and adds 2-6 % false
positives somewhere else (far matches: the dense solver's case), sizes 97..400 (the oracle's Munkres is the slow side there)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mot_amd, orc

budget = float(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
missfp = len(sys.argv) > 3 and int(sys.argv[3]) != 0
rng = np.random.default_rng(seed)
lib = orc.load_oracle()
c = mot_amd.MotContext(max_tracks=1024, max_dets=1024)
t0 = time.time(); n_cases = 0; used = [0, 0, 0]
while time.time() - t0 < budget:
    n = int(rng.integers(97, 401 if missfp else 1025))
    spread = int(rng.choice([2, 4, 8, 16]))
    grid = int(rng.choice([1, 1, 2, 4]))                               # coarse grids -> many equal distances
    area_w, area_h = int(rng.integers(300, 1200)), int(rng.integers(200, 640))
    cx = rng.integers(40, 40 + area_w, n) // grid * grid; cy = rng.integers(40, 40 + area_h, n) // grid * grid
    ndup = int(rng.integers(0, 6))
    for _ in range(ndup):                                              # objects sharing a centroid and class
        a, b = rng.integers(0, n, 2); cx[b] = cx[a]; cy[b] = cy[a]
    typ = np.arange(n) % 3
    for _ in range(ndup):
        a, b = rng.integers(0, n, 2); typ[b] = typ[a]
    trk = [(int(cx[i] + rng.integers(-spread, spread + 1)) - 40, int(cy[i] + rng.integers(-spread, spread + 1)) - 40, int(cy[i]) + 39, int(cx[i]) + 39, int(typ[i]), 0.9) for i in range(n)]
    keep = rng.permutation(n)[: int(rng.integers(max(n - 3, 1), n + 1))] if rng.integers(0, 4) == 0 else rng.permutation(n)
    det = [(int(cx[i] + rng.integers(-2, 3)) - 40, int(cy[i] + rng.integers(-2, 3)) - 40, int(cy[i]) + 39, int(cx[i]) + 39, int(typ[i]), 0.9) for i in keep]
    if missfp:
        kept = [dd for dd in det if rng.integers(0, 100) >= int(rng.integers(2, 7))]
        nfp = max(1, int(len(det) * rng.integers(2, 7) / 100))
        for _ in range(nfp):
            fx, fy = int(rng.integers(40, 1240)), int(rng.integers(40, 680))
            kept.append((fx - 40, fy - 40, fy + 39, fx + 39, int(rng.integers(0, 3)), 0.9))
        det = [kept[i] for i in rng.permutation(len(kept))]
    nT, nD = len(trk), len(det)
    at, ad, cost = c.assign(trk, det)
    d = orc.cost_matrix(lib, trk, det)
    if nT < nD:
        ra, rc = orc.assignment_optimal(lib, d, nT, nD); ok = np.array_equal(at, ra)
    else:
        ra, rc = orc.assignment_optimal(lib, d, nD, nT); ok = np.array_equal(ad, ra)
    st = c.lap_stats()
    used[int(st[15]) % 3] += 1
    if not ok or cost != rc:
        # never overwrite: every failing problem becomes a regression fixture (tests/golden/make_assoc_regressions.py appends it)
        os.makedirs("gpurun_out", exist_ok=True)
        k = 0
        while os.path.exists(f"gpurun_out/assign_stress_fail_{seed}_{k}.npz"):
            k += 1
        np.savez(f"gpurun_out/assign_stress_fail_{seed}_{k}.npz", trk=mot_amd.boxes_array(trk), det=mot_amd.boxes_array(det))
        print("MISMATCH", n, nT, nD, st[:16].tolist(), f"-> gpurun_out/assign_stress_fail_{seed}_{k}.npz"); sys.exit(1)
    n_cases += 1
st = c.lap_stats()
print(f"assign_stress OK: {n_cases} problems, decided by certificate / sparse emulation / dense emulation = {used}; dense solver ran in {int(st[29])} launches, {int(st[30])} of them certified")
