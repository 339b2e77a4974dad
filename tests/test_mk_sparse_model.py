"""CPU checks of the sparse order-exact Munkres emulation's model (oracle/mk_sparse_model.c, a CPU model of csrc/mk_sparse.hip)
against the oracle's dense order-exact Munkres (oracle/mot_oracle.c:orc_assignment_optimal, hungarian.cpp:29-368).

Properties: (1) status == 0 (the a-posteriori check of every non-candidate entry passed) implies assignment == reference
assignment; (2) the BATCHED event loop (what the device runs: up to 64 order-neutral step-3 events per iteration) performs exactly
the primes, step-5 passes and augmentations of the one-event-at-a-time loop and ends with the same assignment."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import orc
from orc import P
from test_lap_model import mm


class MksInfo(C.Structure):
    _fields_ = [("status", C.c_int), ("primes", C.c_long), ("s5", C.c_long), ("aug", C.c_long), ("maxS", C.c_double), ("iters", C.c_long)]


@pytest.fixture(scope="module")
def mks():
    orc.build_oracle()
    so = os.path.join(orc.ORACLE_DIR, "libmk_sparse_model.so")
    if orc.ORACLE_DIR == orc.ORACLE_SRC_DIR and (not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(orc.ORACLE_SRC_DIR, "mk_sparse_model.c"))):
        subprocess.check_call(["make", "-C", orc.ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)
    return C.CDLL(so)


def run(lib, d, nr, nc, batch):
    a = np.full(max(nr, 1), -1, np.int32)
    info = MksInfo()
    d = np.ascontiguousarray(d, np.float64)
    margin = C.c_double(1e-9 * (1.0 + float(d.max()) if d.size else 1.0))
    if batch:
        lib.mks_solve_batched(P(d), nr, nc, 8, margin, batch, P(a), C.byref(info))
    else:
        lib.mks_solve(P(d), nr, nc, 8, margin, P(a), C.byref(info))
    return a[:nr], info


def test_sparse_model_vs_oracle_and_batched_loop(mks):
    lib = orc.load_oracle()
    rng = np.random.default_rng(20261003)
    accepted = batched_shorter = 0
    for it in range(700):
        kind = int(rng.integers(0, 9))
        nr = int(rng.integers(1, 90)); nc = nr + int(rng.integers(0, 12))
        d = mm(rng, nr, nc, kind)
        a0, i0 = run(mks, d, nr, nc, 0)
        for b in (64, 2):
            a1, i1 = run(mks, d, nr, nc, b)
            assert (i0.status, i0.primes, i0.s5, i0.aug) == (i1.status, i1.primes, i1.s5, i1.aug), (it, kind, nr, nc, b)
            assert np.array_equal(a0, a1), (it, kind, nr, nc, b)
            batched_shorter += (b == 64 and i1.iters < i0.primes + i0.s5)
        if i0.status == 0:
            ref, _ = orc.assignment_optimal(lib, d, nr, nc)
            assert np.array_equal(a0, ref), (it, kind, nr, nc)
            accepted += 1
    assert accepted > 100 and batched_shorter > 20, (accepted, batched_shorter)


def run_traced(lib, d, nr, nc, lazy, k=8):
    a = np.full(max(nr, 1), -1, np.int32)
    info = MksInfo()
    tr = np.zeros(8192, np.uint64)
    d = np.ascontiguousarray(d, np.float64)
    margin = C.c_double(1e-9 * (1.0 + float(d.max()) if d.size else 1.0))
    lib.mks_solve_traced(P(d), nr, nc, k, margin, lazy, P(a), C.byref(info), P(tr), len(tr))
    return a[:nr], info, tr


def tracking_like(rng, n, extra=0, dup=4, spread=6):
    """cost matrix (column-major, n x (n + extra)) of a crowded scene: centroid distance / 1280 + class penalty (td.cpp:407-419), a few
    rows sitting on ONE centroid (tied optima), predictions off by up to `spread` px so that many rows want the same column"""
    cx = rng.integers(40, 1240, n + extra).astype(np.float64); cy = rng.integers(40, 680, n + extra).astype(np.float64)
    cls = np.arange(n + extra) % 3
    tx = cx[:n] + rng.integers(-spread, spread + 1, n); ty = cy[:n] + rng.integers(-spread, spread + 1, n); tc = cls[:n].copy()
    for q in range(dup):
        i, j = rng.integers(0, n, 2)
        tx[i], ty[i], tc[i] = tx[j], ty[j], tc[j]
    perm = rng.permutation(n + extra)
    dx, dy, dc = cx[perm], cy[perm], cls[perm]
    d = np.sqrt((tx[:, None] - dx[None, :]) ** 2 + (ty[:, None] - dy[None, :]) ** 2) * (1.0 / 1280.0) + (tc[:, None] != dc[None, :]) * 1.0
    return np.asfortranarray(d).ravel(order="F")


def test_lazy_reset_is_step_for_step_the_full_reset(mks):
    """The lazy reset (only DIRTY components of the zero graph are uncovered and re-grown after an augmentation; what the device runs
    since round 4) against the reference's full reset (hungarian.cpp:324-334): the machine's state -- row covers, column covers, primes,
    stars -- hashed at EVERY step-5 entry must be the same sequence, with the same number of step 5s and augmentations and the same
    assignment; only the number of step-3 events may differ (it must drop on crowded tracking scenes)."""
    rng = np.random.default_rng(20261004)
    ev_full = ev_lazy = 0
    for it in range(500):
        kind = int(rng.integers(0, 9))
        nr = int(rng.integers(1, 120)); nc = nr + int(rng.integers(0, 12))
        d = mm(rng, nr, nc, kind)
        a0, i0, t0 = run_traced(mks, d, nr, nc, 0)
        a1, i1, t1 = run_traced(mks, d, nr, nc, 1)
        assert (i0.status, i0.s5, i0.aug) == (i1.status, i1.s5, i1.aug) and np.array_equal(a0, a1) and np.array_equal(t0, t1), (it, kind, nr, nc)
        assert i1.primes <= i0.primes
    for it in range(12):
        n = int(rng.integers(300, 1025)); extra = int(rng.integers(0, 2)) * int(rng.integers(1, 20))
        d = tracking_like(rng, n, extra, spread=int(rng.integers(6, 31)))
        a0, i0, t0 = run_traced(mks, d, n, n + extra, 0)
        a1, i1, t1 = run_traced(mks, d, n, n + extra, 1)
        assert (i0.status, i0.s5, i0.aug) == (i1.status, i1.s5, i1.aug) and np.array_equal(a0, a1) and np.array_equal(t0, t1), ("tracking", it, n, extra)
        ev_full += i0.primes; ev_lazy += i1.primes
    assert ev_lazy * 2 < ev_full, (ev_full, ev_lazy)
