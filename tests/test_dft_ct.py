"""csrc/dft_ct.h on the host: the column pass of the general forward transform as two short passes (verdict item 6 of round 4, prepared in
round 5 behind -DMOT_FFT_MIXED=1) against numpy, for every line length 8..64 that has a factor 2..5, odd and even bin counts, several
"workgroup sizes".  No GPU: the header is plain C++ over pointers and ints."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    out = os.path.join(tempfile.gettempdir(), f"dft_ct_host_{os.getpid()}.so")
    # MOT_DFT_CT_FLAGS: extra compiler flags (tests/test_sanitizers.py: -fsanitize=address,undefined)
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-ffp-contract=off"] + os.environ.get("MOT_DFT_CT_FLAGS", "").split() +
                          ["-o", out, os.path.join(ROOT, "tests", "dft_ct_host.cpp")])
    l = C.CDLL(out)
    l.ct_small_factor.argtypes = [C.c_int]
    l.ct_cols.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5
    l.ct_cols_a_reversed.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 5
    l.ct_cols_inplace.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4
    l.ct_rows_factor.argtypes = [C.c_int]
    l.ct_forward2d.argtypes = [C.c_void_p] * 5 + [C.c_int] * 6
    l.ct_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 6
    yield l
    os.unlink(out)


def _tw(n):
    j = np.arange(n)
    return np.ascontiguousarray(np.stack([np.cos(2 * np.pi * j / n), np.sin(2 * np.pi * j / n)], axis=1).astype(np.float32))


def test_small_factor_choice(lib):
    for n in range(1, 130):
        f = lib.ct_small_factor(n)
        cands = [q for q in (2, 3, 4, 5) if n % q == 0 and q + n // q < n] if n >= 8 else []
        if not cands: assert f == 0, n
        else: assert f in cands and f + n // f == min(q + n // q for q in cands), n
    assert lib.ct_small_factor(24) == 4 and lib.ct_small_factor(18) == 3 and lib.ct_small_factor(22) == 2 and lib.ct_small_factor(23) == 0 and lib.ct_small_factor(30) == 5


@pytest.mark.parametrize("n", [n for n in range(8, 65) if any(n % q == 0 for q in (2, 3, 4, 5))])
def test_two_step_columns_equal_numpy(lib, n):
    rng = np.random.default_rng(n)
    for N1 in [q for q in (2, 3, 4, 5) if n % q == 0]:                   # every legal split, not only the chosen one
        for fh, nch, nt in ((n // 2 + 1, 3, 64), (7, 2, 512), (10, 1, 17)):
            x = (rng.standard_normal((nch, n, fh)) + 1j * rng.standard_normal((nch, n, fh))).astype(np.complex64)
            ref = np.fft.fft(x.astype(np.complex128), axis=1)
            T = np.ascontiguousarray(x.view(np.float32)); out = np.zeros_like(T); tw = _tw(n)
            lib.ct_cols(T.ctypes.data, out.ctypes.data, tw.ctypes.data, n, N1, fh, nch, nt)
            got = out.view(np.complex64).reshape(nch, n, fh)
            err = np.abs(got - ref).max() / np.abs(ref).max()
            assert err < 2e-6, (n, N1, fh, nch, nt, err)


def test_step_a_is_order_independent(lib):
    """in place by ownership: visiting the work items in another order leaves the same bits"""
    n, N1, fh, nch = 24, 4, 13, 4
    rng = np.random.default_rng(5)
    x = np.ascontiguousarray(rng.standard_normal((nch, n, fh, 2)).astype(np.float32))
    a = x.copy(); b = x.copy(); out = np.zeros_like(x); tw = _tw(n)
    lib.ct_cols(a.ctypes.data, out.ctypes.data, tw.ctypes.data, n, N1, fh, nch, 96)
    lib.ct_cols_a_reversed(b.ctypes.data, tw.ctypes.data, n, N1, fh, nch, 40)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("n", [16, 18, 19 * 2, 21, 22, 24, 25, 27, 30])
def test_in_place_variant_equals_numpy(lib, n):
    """the column pass of dft2_generic_inplace under MOT_FFT_MIXED: step A, step C's sums held back across the barrier, then written over the inputs"""
    rng = np.random.default_rng(100 + n)
    N1 = lib.ct_small_factor(n)
    assert N1
    for fh, nch in ((n // 2 + 1, 3), (9, 2)):
        x = (rng.standard_normal((nch, n, fh)) + 1j * rng.standard_normal((nch, n, fh))).astype(np.complex64)
        ref = np.fft.fft(x.astype(np.complex128), axis=1)
        S = np.ascontiguousarray(x.view(np.float32)); tw = _tw(n)
        lib.ct_cols_inplace(S.ctypes.data, tw.ctypes.data, n, N1, fh, nch)
        got = S.view(np.complex64).reshape(nch, n, fh)
        assert np.abs(got - ref).max() / np.abs(ref).max() < 2e-6, (n, N1, fh)


@pytest.mark.parametrize("n", [n for n in range(8, 65) if any(n % q == 0 for q in (2, 3, 4, 5))])
def test_two_step_rows_equal_numpy(lib, n):
    """real lines -> half spectrum: step A packs the short real DFTs into the line itself, step C folds the twiddles into its sums"""
    rng = np.random.default_rng(1000 + n)
    fh = n // 2 + 1; ldf = 2 * fh
    for N1 in [q for q in (2, 3, 4, 5) if n % q == 0]:
        for lines, nt in ((7, 64), (48, 512), (1, 5)):
            x = rng.standard_normal((lines, n)).astype(np.float32)
            ref = np.fft.rfft(x.astype(np.float64), axis=1)
            F = np.full((lines, ldf), 7.0, np.float32); F[:, :n] = x
            T = np.zeros((lines, fh, 2), np.float32); tw = _tw(n)
            lib.ct_rows(F.ctypes.data, T.ctypes.data, tw.ctypes.data, n, N1, fh, ldf, lines, nt)
            got = T.view(np.complex64).reshape(lines, fh)
            assert np.abs(got - ref).max() / np.abs(ref).max() < 2e-6, (n, N1, lines, nt)
            assert np.all(F[:, n:] == 7.0), "the pad floats of a line are not the pass's to touch"
    f = lib.ct_rows_factor(n)
    assert f in (0, 3, 4, 5) and (f == 0 or n % f == 0)


@pytest.mark.parametrize("hb,wb", [(24, 24), (16, 18), (18, 16), (21, 24), (24, 15), (50, 50), (42, 30), (20, 25)])
def test_forward_2d_as_the_kernel_strings_the_passes(lib, hb, wb):
    """rows then columns through the layouts the kernel uses (feature lines of ldf = 2*fh floats, x slow, y fast; half spectrum along y)"""
    rng = np.random.default_rng(hb * 100 + wb)
    nch, fh = 5, hb // 2 + 1; ldf = 2 * fh
    r1 = lib.ct_rows_factor(hb); c1 = lib.ct_small_factor(wb)
    assert r1 and c1
    x = rng.standard_normal((nch, wb, hb)).astype(np.float32)
    ref = np.fft.fft(np.fft.rfft(x.astype(np.float64), axis=2), axis=1)            # [ch][x'][k]
    B = np.zeros((nch, wb, ldf), np.float32); B[:, :, :hb] = x
    T = np.zeros((nch, wb, fh, 2), np.float32); out = np.zeros_like(T)
    twr, twc = _tw(hb), _tw(wb)                                            # (kept alive across the call)
    lib.ct_forward2d(B.ctypes.data, T.ctypes.data, out.ctypes.data, twr.ctypes.data, twc.ctypes.data, hb, wb, nch, r1, c1, 512)
    got = out.view(np.complex64).reshape(nch, wb, fh)
    assert np.abs(got - ref).max() / np.abs(ref).max() < 3e-6
