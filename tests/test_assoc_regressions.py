"""Every association problem on which the device path ever disagreed with the checker, kept as a fixture
(tests/golden/assoc_regressions.npz; expected output = the reference's own assignmentoptimal, generator:
tests/golden/make_assoc_regressions.py).  CPU: the oracle restatement and the CPU models of the tiers reproduce it.
GPU: the device path reproduces it with the default tiers, with the batched event loop off and with the fast path off."""
import os
import subprocess
import sys

import numpy as np
import pytest

import orc

ROOT = orc.ROOT
FIX = os.path.join(ROOT, "tests", "golden", "assoc_regressions.npz")


def _cases():
    g = np.load(FIX)
    return [(g[f"trk_{k}"], g[f"det_{k}"], g[f"a_{k}"], float(g[f"c_{k}"])) for k in range(int(g["n"]))]


def test_oracle_reproduces_regression_fixtures(oracle):
    cases = _cases()
    assert len(cases) >= 1
    for trk, det, a, c in cases:
        nT, nD = len(trk), len(det)
        nr, nc = (nT, nD) if nT < nD else (nD, nT)
        ao, co = orc.assignment_optimal(oracle, orc.cost_matrix(oracle, trk, det), nr, nc)
        assert np.array_equal(ao, a) and co == c


_CODE = r'''
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import mot_amd
g = np.load(os.path.join("tests", "golden", "assoc_regressions.npz"))
c = mot_amd.MotContext(max_tracks=1024, max_dets=1024)
used = []
for k in range(int(g["n"])):
    trk, det = g[f"trk_{k}"], g[f"det_{k}"]
    for rep in range(3):                                                # state left by one launch must not leak into the next
        at, ad, cost = c.assign(trk, det)
        got = at if len(trk) < len(det) else ad
        assert np.array_equal(got, g[f"a_{k}"]) and cost == float(g[f"c_{k}"]), (k, rep)
    used.append(int(c.lap_stats()[15]))
print("REGRESSIONS_OK", used)
'''


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"MOT_MK_BATCH": "0"}, {"MOT_MK_LAZY": "0"}, {"MOT_LAP_FAST": "0"}, {"MOT_LAP_MIN": "1", "MOT_LAP_DENSE": "1"}],
                         ids=["default", "one_event_loop", "full_reset", "fast_path_off", "dense_solver_forced"])
def test_device_reproduces_regression_fixtures(env):
    out = subprocess.run([sys.executable, "-c", _CODE], cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert "REGRESSIONS_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
