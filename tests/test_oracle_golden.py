"""Pins the oracle (oracle/mot_oracle.c) against golden vectors produced by the
reference's own sources (tests/golden/make_golden.py).  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

import orc
from orc import P, BBox

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def test_sse_rcp_rsqrt_model(oracle):
    g = load("sse_approx.npz")
    x = g["x"].view(np.float32)
    rc = np.array([oracle.orc_sse_rcp(float(v)) for v in x[:4000]], np.float32)
    # ctypes float round trip drops NaN payloads; compare non-NaN bit patterns
    ref = g["rcp"][:4000].view(np.float32)
    ok = ~np.isnan(ref)
    assert np.array_equal(rc.view(np.uint32)[ok], ref.view(np.uint32)[ok])
    rs = np.array([oracle.orc_sse_rsqrt(float(v)) for v in x[:4000]], np.float32)
    ref = g["rsqrt"][:4000].view(np.float32)
    ok = ~np.isnan(ref)
    assert np.array_equal(rs.view(np.uint32)[ok], ref.view(np.uint32)[ok])
    assert np.all(np.isnan(rs[~ok]))


def test_acos_table(oracle):
    ref = load("acos_table.npz")["table"]
    p = oracle.orc_acos_table()
    tab = np.ctypeslib.as_array(C.cast(C.addressof(p.contents) - 4 * 10010, C.POINTER(C.c_float)), shape=(20020,))
    assert np.array_equal(tab.view(np.uint32), ref.view(np.uint32))


def test_bin_thresholds_match_reference_table():
    """the 2x9 integer thresholds compiled into the HIP kernel reproduce the reference's LUT + quantiser"""
    import re
    inc = open(os.path.join(orc.ROOT, "multiple-object-tracking_amd", "csrc", "bin_thresholds.inc")).read()
    thr = [list(map(int, re.search(r"MOT_BIN_THR%d \{([^}]*)\}" % f, inc).group(1).split(","))) for f in (0, 1)]
    tab = load("acos_table.npz")["table"]
    PI = np.float32(3.14159265)
    oMult = np.float32(np.float32(18) / (np.float32(2) * PI))
    idx = np.arange(-10010, 10010)
    for flag, top in ((0, 9), (1, 18)):
        O = (tab + (PI if flag else np.float32(0))).astype(np.float32)
        o0 = ((O * oMult).astype(np.float32) + np.float32(0.5)).astype(np.float32).astype(np.int32)
        rec = top - sum((idx >= t).astype(np.int32) for t in thr[flag])
        assert np.array_equal(rec, o0)


@pytest.mark.parametrize("name,mode", [("fhog_cases.npz", 0), ("fhog_cases_exact.npz", 1)])
def test_fhog_bit_exact(oracle, name, mode):
    """mode 0: the reference as it is (rcpps / rsqrtps, libhog/sse.hpp:40-41); mode 1: its exact-math flavour (the same sources compiled with
    correctly rounded 1 / x and 1 / sqrt(x), oracle/ref_exact_sse.h) -- SURVEY 8c asks for fixtures of both"""
    g = load(name)
    for i in range(int(g["n"])):
        h, w = map(int, g[f"c{i}_hw"])
        I = np.ascontiguousarray(g[f"c{i}_I"])
        M = np.zeros(h * w, np.float32); O = np.zeros(h * w, np.float32)
        oracle.orc_grad_mag(P(I), P(M), P(O), h, w, mode)
        assert np.array_equal(M.view(np.uint32), g[f"c{i}_M"].view(np.uint32)), f"case {i} M"
        assert np.array_equal(O.view(np.uint32), g[f"c{i}_O"].view(np.uint32)), f"case {i} O"
        H = orc.fhog(oracle, I, h, w, mode)
        assert np.array_equal(H.view(np.uint32), g[f"c{i}_H"].view(np.uint32)), f"case {i} H"
        assert not H[31 * (h // 4) * (w // 4):].any()


def test_crop_resize_bit_exact(oracle):
    g = load("crop_cases.npz")
    frame = np.random.default_rng(int(g["seed"])).integers(0, 256, size=(720, 1280, 3), dtype=np.uint8)
    for i in range(int(g["n"])):
        l, t, r, b, rows, cols = map(int, g[f"c{i}_box"])
        gray = np.zeros((b - t + 1) * (r - l + 1), np.float32)
        oracle.orc_rgb2gray(P(gray), P(frame), l, t, r, b)
        assert np.array_equal(gray.view(np.uint32), g[f"c{i}_gray"].view(np.uint32))
        patch = orc.crop_patch(oracle, frame, (l, t, b, r), rows, cols)
        assert np.array_equal(patch.view(np.uint32), g[f"c{i}_patch"].view(np.uint32))


def test_kalman(oracle):
    g = load("kalman_cases.npz")
    for tr in range(g["box0"].shape[0]):
        l, t, b, r = map(int, g["box0"][tr])
        b0 = BBox(l, t, b, r, 0, 0.9)
        k = C.c_void_p(oracle.orc_kalman_new(C.byref(b0)))
        for s in range(g["z"].shape[1]):
            p = BBox()
            oracle.orc_kalman_predict(k, C.byref(p))
            assert (p.l, p.t, p.b, p.r) == tuple(int(v) for v in g["pred"][tr, s])
            zl, zt, zb, zr = map(int, g["z"][tr, s])
            z = BBox(zl, zt, zb, zr, 0, 0.9)
            oracle.orc_kalman_update(k, C.byref(z))
            x = np.zeros(6); Pm = np.zeros(36)
            oracle.orc_kalman_get_state(k, P(x), P(Pm))
            # tolerance: float64 6x6 products summed in a different order than the reference's BLAS
            np.testing.assert_allclose(x, g["x"][tr, s], rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(Pm, g["P"][tr, s], rtol=1e-12, atol=1e-9)
        oracle.orc_kalman_delete(k)


def _munkres_matrix(rng, nr, nc, kind):
    import importlib.util
    spec = importlib.util.spec_from_file_location("mg", os.path.join(G, "make_golden.py"))
    # the generator lives next to the fixtures; re-implement the recipe here to avoid importing it
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
    else:
        d = np.round(rng.uniform(0, 1, size=nr * nc) * 8) / 8.0
    return np.ascontiguousarray(d, np.float64)


def test_munkres_bit_exact(oracle):
    g = load("munkres_cases.npz")
    for i in range(int(g["n"])):
        nr, nc, _ = map(int, g[f"m{i}_shape"])
        a, c = orc.assignment_optimal(oracle, g[f"m{i}_d"], nr, nc)
        assert np.array_equal(a, g[f"m{i}_a"]), f"matrix {i}"
        assert c == float(g[f"m{i}_c"]), f"matrix {i} cost"
    for i in range(int(g["nbig"])):
        nn, kind, seed = map(int, g[f"big{i}_spec"])
        d = _munkres_matrix(np.random.default_rng(seed), nn, nn, kind)
        a, c = orc.assignment_optimal(oracle, d, nn, nn)
        assert np.array_equal(a, g[f"big{i}_a"])
        assert c == float(g[f"big{i}_c"])


@pytest.mark.parametrize("name", ["kcf_seq_80.npz", "kcf_seq_64.npz", "kcf_seq_148.npz", "kcf_seq_200.npz"])
def test_kcf_sequence(oracle, name):
    g = load(name)
    S = int(g["S"]); fr = S // 4; nf = fr * fr; nh = fr * (fr // 2 + 1)
    l, t, b, r, ty = map(int, g["box0"])
    b0 = BBox(l, t, b, r, ty, 0.9)
    k = C.c_void_p(oracle.orc_kcf_new(C.byref(b0), 0))
    assert np.array_equal(orc.arr(oracle.orc_kcf_labels(k), nf), g["labels"])
    assert np.array_equal(orc.arr(oracle.orc_kcf_coswin(k), nf), g["coswin"])
    # FFT results: the reference's FFT library (FFTW / MKL) rounds differently from the oracle's DFT -> 1e-6 relative
    np.testing.assert_allclose(orc.arr(oracle.orc_kcf_yf(k), nh * 2), g["yf"], rtol=0, atol=1e-6 * np.abs(g["yf"]).max())
    p = np.ascontiguousarray(g["p_init"])
    oracle.orc_kcf_update(k, P(p), C.byref(b0))
    assert np.array_equal(orc.arr(oracle.orc_kcf_features(k), 31 * nf).view(np.uint32), g["feat_init"].view(np.uint32))
    np.testing.assert_allclose(orc.arr(oracle.orc_kcf_alpha(k), nh), g["alpha_init"], rtol=0, atol=2e-6 * np.abs(g["alpha_init"]).max())
    xm = orc.arr(oracle.orc_kcf_xm(k), 31 * nh * 2)[::37]
    np.testing.assert_allclose(xm, g["xm_init_sample"], rtol=0, atol=1e-6 * np.abs(g["xm_init_sample"]).max())
    for s in range(int(g["steps"])):
        p = np.ascontiguousarray(g[f"s{s}_patch"])
        pb = BBox()
        oracle.orc_kcf_predict(k, P(p), C.byref(pb))
        resp = orc.arr(oracle.orc_kcf_response(k), nf)
        ref = g[f"s{s}_resp"]
        assert resp.argmax() == ref.argmax()
        assert abs(resp.max() - ref.max()) <= 1e-5 * abs(ref.max())          # restatement vs oracle fixtures <= 1e-5
        np.testing.assert_allclose(resp, ref, rtol=0, atol=1e-5 * np.abs(ref).max())
        assert (pb.l, pb.t, pb.b, pb.r, pb.type) == tuple(int(v) for v in g[f"s{s}_pred"])
        ul, ut, ub, ur, uty = map(int, g[f"s{s}_ubox"])
        nb = BBox(ul, ut, ub, ur, uty, 0.9)
        oracle.orc_kcf_update(k, P(p), C.byref(nb))
        np.testing.assert_allclose(orc.arr(oracle.orc_kcf_alpha(k), nh), g[f"s{s}_alpha"], rtol=0, atol=5e-6 * np.abs(g[f"s{s}_alpha"]).max())
    xm = orc.arr(oracle.orc_kcf_xm(k), 31 * nh * 2)[::37]
    np.testing.assert_allclose(xm, g["xm_final_sample"], rtol=0, atol=2e-6 * np.abs(g["xm_final_sample"]).max())
    oracle.orc_kcf_delete(k)


def _scene(spec):
    import mot_amd  # noqa: F401  (registers the package)
    from multiple_object_tracking_amd import synth
    n, size, sid, miss, fp, nframes = map(int, spec[:6])
    ds = (int(spec[6]), int(spec[7])) if len(spec) > 6 else None
    return synth.Scene(n, size, stream_id=sid, det_sizes=ds, miss_pct=miss, fp_pct=fp), nframes


@pytest.mark.parametrize("name,kind", [("frameloop_kalman.npz", 1), ("frameloop_kcf.npz", 0), ("frameloop_kcf_multiscale.npz", 0)])
def test_frame_loop_trace(oracle, name, kind):
    """orc_mot_step (td.cpp:306-748 restated) against a trace obtained by driving the reference's own
    tracker_* / assignmentoptimal / rgb2Gray / bilinearInterpolationGray through the same loop."""
    g = load(name)
    scene, nframes = _scene(g["spec"])
    m = orc.OracleMot(oracle, kind, 0, 256)
    for f, (frame, dets) in enumerate(scene.frames(nframes)):
        out = m.step(frame, dets)
        pred = np.stack([out["predicted"][k] for k in ("l", "t", "b", "r", "type")], axis=1).reshape(-1, 5)
        assert np.array_equal(pred, g[f"f{f}_pred"]), f"frame {f} predicted"
        assert np.array_equal(out["assigned"], g[f"f{f}_assigned"]), f"frame {f} assignment"
        live = np.stack([out["live"][k] for k in ("l", "t", "b", "r", "type")], axis=1).reshape(-1, 5)
        assert np.array_equal(live, g[f"f{f}_live"]), f"frame {f} live"
        assert np.array_equal(out["tids"].astype(np.int32), g[f"f{f}_tids"])
    m.close()
