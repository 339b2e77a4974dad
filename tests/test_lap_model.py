"""CPU fuzz of the assignment fast path's certificate (oracle/lap_model.c, a CPU model of csrc/lap_kernels.hip)
against the oracle's order-exact Munkres (oracle/mot_oracle.c:orc_assignment_optimal, hungarian.cpp:29-368).

Property: status == 0 ("optimum unique with margin") must imply assignment == reference assignment.  The
uncertified outcomes (ties, solver gave up) are the cases the device path hands to the order-exact emulation."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import orc
from orc import P

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Info(C.Structure):
    _fields_ = [("status", C.c_int), ("rounds", C.c_int), ("free0", C.c_int), ("searches", C.c_int), ("commits", C.c_int),
                ("nedges", C.c_int), ("ncyclic", C.c_int), ("hard", C.c_int), ("hard_scans", C.c_int), ("eps", C.c_double), ("gamma", C.c_double), ("cmax", C.c_double)]


@pytest.fixture(scope="module")
def model():
    orc.build_oracle()
    so = os.path.join(orc.ORACLE_DIR, "liblap_model.so")
    if orc.ORACLE_DIR == orc.ORACLE_SRC_DIR and (not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(orc.ORACLE_SRC_DIR, "lap_model.c"))):
        subprocess.check_call(["make", "-C", orc.ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)
    lib = C.CDLL(so)
    lib.lapm_solve.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return lib


def solve(lib, d, nr, nc, K=8, S=128):
    a = np.full(max(nr, 1), -1, np.int32)
    info = Info()
    st = lib.lapm_solve(P(np.ascontiguousarray(d, np.float64)), nr, nc, K, S, P(a), C.byref(info))
    return st, a[:nr], info


def mm(rng, nr, nc, kind):
    """the matrix families of tests/test_gpu_parity.py:_mm plus tracking-like ones"""
    if kind == 0:
        d = rng.uniform(0, 1, size=nr * nc)
    elif kind == 1:
        d = rng.integers(0, 6, size=nr * nc).astype(np.float64)
    elif kind == 2:
        d = rng.integers(0, 40, size=nr * nc) / 1280.0 + (rng.integers(0, 3, size=nr * nc) == 0) * 1.0
    elif kind == 3:
        d = np.sqrt(rng.integers(0, 50, size=nr * nc).astype(np.float64)) * (1.0 / 1280)
    elif kind == 4:
        d = np.full(nr * nc, 0.25)
    elif kind == 5:
        d = np.round(rng.uniform(0, 1, size=nr * nc) * 8) / 8.0
    else:
        # td.cpp:386-457 costs of a crowded scene on a coarse pixel grid (duplicated centroids -> tied optima)
        g = 4 if kind == 6 else 1
        span = 40 * int(np.sqrt(max(nr, nc))) // (2 if kind == 8 else 1) + 8
        cx = rng.integers(0, span, size=nc) * g; cy = rng.integers(0, span, size=nc) * g; ty = rng.integers(0, 3, size=nc)
        own = rng.permutation(nc)[:nr]
        rx = cx[own] + rng.integers(-3, 4, size=nr) * g; ry = cy[own] + rng.integers(-3, 4, size=nr) * g; rt = ty[own]
        dx = rx[:, None] - cx[None, :]; dy = ry[:, None] - cy[None, :]
        m = np.sqrt((dx * dx + dy * dy).astype(np.float64)) * (1.0 / 1280) + (rt[:, None] != ty[None, :]) * 1.0
        d = m.T.reshape(-1)                                          # column-major
    return np.ascontiguousarray(d, np.float64)


def test_certified_implies_reference_assignment(model, oracle):
    rng = np.random.default_rng(2024)
    n_cert = n_unc = 0
    by_status = {}
    for trial in range(2500):
        kind = trial % 9
        nr = int(rng.integers(1, 90)); nc = int(rng.integers(nr, 100))
        if trial % 3 == 0:
            nc = nr
        d = mm(rng, nr, nc, kind)
        st, a, info = solve(model, d, nr, nc, K=int(rng.integers(2, 13)), S=int(rng.integers(1, 64)))
        by_status[st] = by_status.get(st, 0) + 1
        if st == 0:
            ra, _ = orc.assignment_optimal(oracle, d, nr, nc)
            assert np.array_equal(a, ra), f"trial {trial} kind {kind} {nr}x{nc}: certified but differs from the reference"
            n_cert += 1
        else:
            n_unc += 1
    assert n_cert > 500 and n_unc > 100, by_status               # both outcomes exercised


def test_near_ties_never_certify_wrongly(model, oracle):
    """tied optima broken by perturbations from 1e-16 to 1e-5: below the margin the model must refuse, above it the
    reference follows the perturbation -- either way certified == reference"""
    rng = np.random.default_rng(5)
    seen = set()
    for trial in range(600):
        n = int(rng.integers(4, 40))
        d = mm(rng, n, n, 6 + trial % 3).reshape(n, n).T.copy()      # row-major view [r, c]
        # duplicate one column (two tracks on the same centroid) and perturb one copy
        j1, j2 = rng.choice(n, 2, replace=False)
        d[:, j2] = d[:, j1]
        mag = 10.0 ** (-int(rng.integers(5, 17)))
        d[:, j2] += mag * rng.uniform(0.5, 1.0, size=n) * (rng.integers(0, 2, size=n))
        dm = np.ascontiguousarray(d.T.reshape(-1))
        st, a, info = solve(model, dm, n, n)
        seen.add(st)
        if st == 0:
            ra, _ = orc.assignment_optimal(oracle, dm, n, n)
            assert np.array_equal(a, ra), f"trial {trial}: perturbation {mag:g}"
    assert 0 in seen and 4 in seen


def test_golden_munkres_cases(model):
    g = np.load(os.path.join(G, "munkres_cases.npz"))
    n_cert = 0
    for i in range(int(g["n"])):
        nr, nc, _ = map(int, g[f"m{i}_shape"])
        if nr > nc:
            continue
        st, a, info = solve(model, g[f"m{i}_d"], nr, nc)
        if st == 0:
            assert np.array_equal(a, g[f"m{i}_a"]), f"golden matrix {i}"
            n_cert += 1
    assert n_cert > 5


@pytest.mark.parametrize("n", [256, 1024])
def test_large_tracking_scene(model, oracle, n):
    rng = np.random.default_rng(n)
    cx = rng.integers(0, 1200, size=n); cy = rng.integers(0, 640, size=n)
    tx = cx + rng.integers(-6, 7, size=n); ty = cy + rng.integers(-6, 7, size=n)
    perm = rng.permutation(n)
    dx = (cx[perm] + rng.integers(-2, 3, size=n))[:, None] - tx[None, :]
    dy = (cy[perm] + rng.integers(-2, 3, size=n))[:, None] - ty[None, :]
    m = np.sqrt((dx * dx + dy * dy).astype(np.float64)) * (1.0 / 1280) + ((perm % 3)[:, None] != (np.arange(n) % 3)[None, :]) * 1.0
    d = np.ascontiguousarray(m.T.reshape(-1))
    st, a, info = solve(model, d, n, n)
    ra, _ = orc.assignment_optimal(oracle, d, n, n)
    if st == 0:
        assert np.array_equal(a, ra)
    else:
        assert st == 4, f"status {st}"                               # only a genuine tie may refuse a tracking scene


def test_dense_solver_on_miss_and_false_positive_scenes(model):
    """scenes with missed detections AND false positives force far matches: the sparse solver gives up or its prices fail the dense
    check, the dense Jonker-Volgenant solver (CPU statement of csrc/lap_dense.hip) takes over, and its result passes through the same
    dual check and uniqueness certificate -- certified must still imply equal to the oracle's Munkres"""
    lib = orc.load_oracle()
    rng = np.random.default_rng(77)
    dense_certified = dense_ran = 0
    for it in range(250):
        n = int(rng.integers(20, 140))
        cx = rng.integers(40, 1240, n); cy = rng.integers(40, 680, n); typ = np.arange(n) % 3
        trk = [(int(cx[i] + rng.integers(-4, 5)) - 40, int(cy[i] + rng.integers(-4, 5)) - 40, int(cy[i]) + 39, int(cx[i]) + 39, int(typ[i]), 0.9) for i in range(n)]
        det = [(int(cx[i] + rng.integers(-2, 3)) - 40, int(cy[i] + rng.integers(-2, 3)) - 40, int(cy[i]) + 39, int(cx[i]) + 39, int(typ[i]), 0.9)
               for i in rng.permutation(n) if rng.integers(0, 100) >= 6]
        for _ in range(max(1, n // 20)):
            fx, fy = int(rng.integers(40, 1240)), int(rng.integers(40, 680))
            det.append((fx - 40, fy - 40, fy + 39, fx + 39, int(rng.integers(0, 3)), 0.9))
        nT, nD = len(trk), len(det)
        d = orc.cost_matrix(lib, trk, det)
        nr, nc = (nT, nD) if nT < nD else (nD, nT)                    # rows = the smaller side (td.cpp:388, 462-465)
        st, a, info = solve(model, d, nr, nc)
        dense_ran += info.hard > 0
        if st == 0:
            ref, _ = orc.assignment_optimal(lib, d, nr, nc)
            assert np.array_equal(a, ref), (it, nr, nc, info.hard, info.hard_scans)
            dense_certified += info.hard > 0
    assert dense_ran > 150 and dense_certified > 100, (dense_ran, dense_certified)
