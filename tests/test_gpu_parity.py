"""GPU parity tests: the HIP path (through the C ABI, libmot_amd.so) against the
oracle on the same seeded inputs and against the golden fixtures.

Bars (BASELINE.json north_star / SURVEY 8d):
  * FHOG channels, crops, orientation bins: bit-exact
  * Munkres assignment indices and cost, association cost matrix: bit-exact
  * predicted integer boxes, response arg-max: equal
  * KCF response-map peak: <= 1e-4 relative
  * Kalman x, P: <= 1e-12 relative
"""
import ctypes as C
import os

import numpy as np
import pytest

import orc
from orc import P, BBox

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PEAK_RTOL = 1e-4      # north_star: "KCF response-map peak within 1e-4 relative"
KALMAN_RTOL = 1e-12


def load(name):
    return np.load(os.path.join(G, name))


def boxes_to_np(b):
    return np.stack([b[k] for k in ("l", "t", "b", "r", "type")], axis=1).reshape(-1, 5)


@pytest.fixture(scope="module")
def kcf_ctx(mot):
    c = mot.MotContext(tracker_kind=mot.TRACKER_KCF, max_tracks=256, max_dets=256)
    yield c
    c.close()


def _patch(rng, h, w, kind):
    if kind == 0:
        I = rng.integers(0, 256, size=(w, h)).astype(np.float32)
    elif kind == 1:
        yy, xx = np.meshgrid(np.arange(h), np.arange(w))
        I = (128 + 60 * np.sin(xx * 0.1) * np.cos(yy * 0.07) + rng.uniform(-10, 10, size=(w, h))).astype(np.float32)
    else:
        I = np.full((w, h), 50.0, np.float32)
        I[:, h // 3:] = 51.0
    return np.ascontiguousarray(I)


# ---------------------------------------------------------------- FHOG ------
@pytest.mark.parametrize("name,mode", [("fhog_cases.npz", 0), ("fhog_cases_exact.npz", 1)])
def test_fhog_golden_bit_exact(mot, name, mode):
    """both flavours of the reference (as is / correctly rounded 1 / x and 1 / sqrt(x): oracle/ref_exact_sse.h), bit for bit"""
    g = load(name)
    c = mot.MotContext(fhog_mode=mode, max_tracks=4, max_dets=4)
    for i in range(int(g["n"])):
        h, w = map(int, g[f"c{i}_hw"])
        H = c.fhog_extract(g[f"c{i}_I"], h, w)
        assert np.array_equal(H.view(np.uint32), g[f"c{i}_H"].view(np.uint32)), f"golden FHOG case {i} ({h}x{w})"
    c.close()


@pytest.mark.parametrize("h,w", [(80, 80), (64, 64), (96, 48), (40, 120), (148, 148), (150, 150), (200, 120), (9, 8), (83, 77)])
@pytest.mark.parametrize("mode", [0, 1])
def test_fhog_vs_oracle_bit_exact(mot, oracle, h, w, mode):
    c = mot.MotContext(fhog_mode=mode, max_tracks=4, max_dets=4)
    rng = np.random.default_rng(h * 1000 + w + mode)
    for kind in range(3):
        I = _patch(rng, h, w, kind)
        H = c.fhog_extract(I, h, w)
        ref = orc.fhog(oracle, I, h, w, mode)
        assert np.array_equal(H.view(np.uint32), ref.view(np.uint32)), f"{h}x{w} kind {kind} mode {mode}: max diff {np.abs(H - ref).max()}"
    c.close()


def test_crop_resize_bit_exact(kcf_ctx, oracle):
    g = load("crop_cases.npz")
    frame = np.random.default_rng(int(g["seed"])).integers(0, 256, size=(720, 1280, 3), dtype=np.uint8)
    kcf_ctx.frame_upload(frame)
    for i in range(int(g["n"])):
        l, t, r, b, rows, cols = map(int, g[f"c{i}_box"])
        got = kcf_ctx.crop_patch((l, t, b, r), rows, cols)
        assert np.array_equal(got.view(np.uint32), g[f"c{i}_patch"].view(np.uint32)), f"crop case {i}"
    rng = np.random.default_rng(5)
    for _ in range(12):   # random boxes incl. non-square and tiny sources
        l, t = int(rng.integers(0, 1100)), int(rng.integers(0, 600))
        ws, hs = int(rng.integers(3, 170)), int(rng.integers(3, 110))
        rows, cols = int(rng.integers(16, 100)), int(rng.integers(16, 100))
        box = (l, t, t + hs - 1, l + ws - 1)
        got = kcf_ctx.crop_patch(box, rows, cols)
        ref = orc.crop_patch(oracle, frame, box, rows, cols)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"box {box} -> {rows}x{cols}"


# ---------------------------------------------------------------- KCF -------
@pytest.mark.parametrize("name", ["kcf_seq_80.npz", "kcf_seq_64.npz", "kcf_seq_148.npz", "kcf_seq_200.npz"])
@pytest.mark.parametrize("fft_mode", [0, 1])
def test_kcf_sequence_golden(mot, name, fft_mode):
    """tracker_new / tracker_update / tracker_predict semantics with caller-supplied patches
    (trackers/kcf.cpp:455-491) against outputs recorded from the reference."""
    g = load(name)
    S = int(g["S"])
    c = mot.MotContext(max_tracks=8, max_dets=8, fft_mode=fft_mode)
    l, t, b, r, ty = map(int, g["box0"])
    ids = c.tracks_new([(l, t, b, r, ty, 0.9)], first_update=False)
    c.update_batch_patches(ids, [g["p_init"]], [(l, t, b, r, ty, 0.9)])
    xm, alpha = c.get_model(ids[0])
    np.testing.assert_allclose(alpha, g["alpha_init"], rtol=0, atol=1e-5 * np.abs(g["alpha_init"]).max())
    np.testing.assert_allclose(xm[::37], g["xm_init_sample"], rtol=0, atol=1e-5 * np.abs(g["xm_init_sample"]).max())
    for s in range(int(g["steps"])):
        pred = c.predict_batch_patches(ids, [g[f"s{s}_patch"]])
        resp = c.get_response(ids[0])
        ref = g[f"s{s}_resp"]
        assert resp.argmax() == ref.argmax(), f"step {s} arg-max"
        assert abs(resp.max() - ref.max()) <= PEAK_RTOL * abs(ref.max()), f"step {s} peak {resp.max()} vs {ref.max()}"
        assert tuple(boxes_to_np(pred)[0]) == tuple(int(v) for v in g[f"s{s}_pred"]), f"step {s} predicted box"
        ul, ut, ub, ur, uty = map(int, g[f"s{s}_ubox"])
        c.update_batch_patches(ids, [g[f"s{s}_patch"]], [(ul, ut, ub, ur, uty, 0.9)])
        _, alpha = c.get_model(ids[0])
        np.testing.assert_allclose(alpha, g[f"s{s}_alpha"], rtol=0, atol=2e-5 * np.abs(g[f"s{s}_alpha"]).max())
    xm, _ = c.get_model(ids[0])
    np.testing.assert_allclose(xm[::37], g["xm_final_sample"], rtol=0, atol=2e-5 * np.abs(g["xm_final_sample"]).max())
    c.close()


def test_back_to_back_patch_updates_through_the_staging_ring(mot, oracle):
    """Round 5: a per-object update with a caller patch (tracker_update, kcf.cpp:455-476) returns with its launch queued and stages through one of
    two pinned halves.  Seven tracks updated back to back, one call each, nothing synchronised in between and the caller's patch buffer
    overwritten right after every call, for five rounds with different patches; then one predict per track.  Models and responses
    against the oracle: a half reused too early, or a patch read after the call returned, shows up here."""
    rng = np.random.default_rng(11)
    n, rounds = 7, 5
    c = mot.MotContext(max_tracks=8, max_dets=8)
    boxes = [(40 + 90 * i, 60 + 7 * i, 60 + 7 * i + 79, 40 + 90 * i + 79, i % 3, 0.9) for i in range(n)]
    ids = c.tracks_new(boxes, first_update=False)
    oks = [C.c_void_p(oracle.orc_kcf_new(P(orc.boxes_array([b])), 0)) for b in boxes]
    scratch = np.zeros(6400, np.float32)                                # the caller's ONE patch buffer, as td.cpp's grayImage is per tracker but reused per frame
    for r in range(rounds):
        pats = [np.ascontiguousarray(rng.integers(0, 256, 6400).astype(np.float32)) for _ in range(n)]
        for i in range(n):
            scratch[:] = pats[i]
            c.update_batch_patches([ids[i]], [scratch], [boxes[i]])
            scratch[:] = -1.0                                           # overwritten at once: the library must have copied it
        for i, k in enumerate(oks):
            oracle.orc_kcf_update(k, P(pats[i]), P(orc.boxes_array([boxes[i]])))
    probe = [np.ascontiguousarray(rng.integers(0, 256, 6400).astype(np.float32)) for _ in range(n)]
    for i, k in enumerate(oks):
        pred = c.predict_batch_patches([ids[i]], [probe[i]])
        pb = BBox(); oracle.orc_kcf_predict(k, P(probe[i]), C.byref(pb))
        assert tuple(boxes_to_np(pred)[0])[:4] == (pb.l, pb.t, pb.b, pb.r), f"track {i}: predicted box"
        resp = c.get_response(ids[i]); ref = orc.arr(oracle.orc_kcf_response(k), 400)
        assert resp.argmax() == ref.argmax(), f"track {i}: arg-max"
        assert abs(resp.max() - ref.max()) <= PEAK_RTOL * abs(ref.max()), f"track {i}: peak {resp.max()} vs {ref.max()}"
        _, alpha = c.get_model(ids[i]); ra = orc.arr(oracle.orc_kcf_alpha(k), alpha.size)
        np.testing.assert_allclose(alpha, ra, rtol=0, atol=2e-5 * np.abs(ra).max())
    for k in oks:
        oracle.orc_kcf_delete(k)
    c.close()


def test_mixed_template_sizes_with_caller_patches_in_one_batch(mot, oracle):
    """Round-5 advisor finding (mot_ctx.hip, run_batch): the groups of a batch that mixes template sizes were staged at item index x the group's OWN
    patch size, so a later group with the smaller template landed inside the earlier group's patches -- which, on the zero-copy path, the earlier
    group's kernel is still reading over PCIe.  Three 96 x 96 and three 64 x 64 tracks (the larger pool is created first, so its group comes first),
    ids interleaved, ONE batch of six with caller patches: updates, then a predict, both through the zero-copy ring (<= 8 items) and through the
    staged path (a batch of 12 > 8); models and responses against the oracle (kcf.cpp:455-476 per object)."""
    rng = np.random.default_rng(23)
    for n_each in (3, 6):                                               # 6 items: zero-copy ring; 12 items: staged copies
        c = mot.MotContext(max_tracks=16, max_dets=16)
        sizes = [96] * n_each + [64] * n_each
        boxes = [(20 + 100 * (i % 12), 30 + 110 * (i // 12), 30 + 110 * (i // 12) + s - 1, 20 + 100 * (i % 12) + s - 1, i % 3, 0.9) for i, s in enumerate(sizes)]
        ids = c.tracks_new(boxes, first_update=False)
        oks = [C.c_void_p(oracle.orc_kcf_new(P(orc.boxes_array([b])), 0)) for b in boxes]
        order = [j for pair in zip(range(n_each), range(n_each, 2 * n_each)) for j in pair]     # 96, 64, 96, 64, ...
        for r in range(3):
            pats = [np.ascontiguousarray(rng.integers(0, 256, s * s).astype(np.float32)) for s in sizes]
            c.update_batch_patches([ids[j] for j in order], [pats[j] for j in order], [boxes[j] for j in order])
            for j, k in enumerate(oks):
                oracle.orc_kcf_update(k, P(pats[j]), P(orc.boxes_array([boxes[j]])))
        probe = [np.ascontiguousarray(rng.integers(0, 256, s * s).astype(np.float32)) for s in sizes]
        pred = c.predict_batch_patches([ids[j] for j in order], [probe[j] for j in order])
        for q, j in enumerate(order):
            k = oks[j]; nb = (sizes[j] // 4) ** 2
            pb = BBox(); oracle.orc_kcf_predict(k, P(probe[j]), C.byref(pb))
            assert tuple(boxes_to_np(pred[q:q + 1])[0])[:4] == (pb.l, pb.t, pb.b, pb.r), f"{2 * n_each} items, track {j}: predicted box"
            resp = c.get_response(ids[j]); ref = orc.arr(oracle.orc_kcf_response(k), nb)
            assert resp.argmax() == ref.argmax(), f"track {j}: arg-max"
            assert abs(resp.max() - ref.max()) <= PEAK_RTOL * abs(ref.max()), f"track {j}: peak {resp.max()} vs {ref.max()}"
            _, alpha = c.get_model(ids[j]); ra = orc.arr(oracle.orc_kcf_alpha(k), alpha.size)
            np.testing.assert_allclose(alpha, ra, rtol=0, atol=2e-5 * np.abs(ra).max())
        for k in oks:
            oracle.orc_kcf_delete(k)
        c.close()


@pytest.mark.parametrize("rows,cols", [(120, 164), (164, 124), (148, 100), (100, 156)])
def test_kcf_nonsquare_templates_vs_oracle(mot, oracle, rows, cols):
    """non-square templates, most of them beyond the LDS limit (HBM-slab kernels, MFMA DFT with hb != wb): predict / update of a few
    tracks from the bound frame against the oracle -- equal arg-max and boxes, response peak within 1e-4 (kcf.cpp:148-152 freezes
    rows x cols per track; td.cpp:344-384, 512-582)"""
    from multiple_object_tracking_amd import synth
    scene = synth.Scene(6, 80, stream_id=9)
    frames = list(scene.frames(4))
    frame0, d80 = frames[0]
    dets0 = []
    for d in d80:                                                      # same centres, rows x cols boxes inside the frame
        cy, cx = (d[1] + d[2]) // 2, (d[0] + d[3]) // 2
        t = min(max(cy - rows // 2, 0), 720 - rows); l = min(max(cx - cols // 2, 0), 1280 - cols)
        dets0.append((l, t, t + rows - 1, l + cols - 1, d[4], d[5]))
    c = mot.MotContext(max_tracks=8, max_dets=8)
    c.frame_upload(frame0)
    ids = c.tracks_new(dets0)
    nb = (rows // 4) * (cols // 4)
    oks = []
    for d in dets0:
        b = orc.boxes_array([d])
        k = C.c_void_p(oracle.orc_kcf_new(P(b), 0))
        oracle.orc_kcf_update(k, P(orc.crop_patch(oracle, frame0, d, rows, cols)), P(b))
        oks.append(k)
    boxes = [tuple(d) for d in dets0]
    for frame, _ in frames[1:]:
        c.frame_upload(frame)
        pred = c.predict_batch(ids, clamp=True)
        for i, k in enumerate(oks):
            pb = BBox()
            oracle.orc_kcf_predict(k, P(orc.crop_patch(oracle, frame, boxes[i], rows, cols)), C.byref(pb))
            exp = (min(max(pb.l, 0), 1279), min(max(pb.t, 0), 719), min(max(pb.b, 0), 719), min(max(pb.r, 0), 1279), pb.type)
            assert tuple(boxes_to_np(pred[i:i + 1])[0]) == exp, f"track {i}"
            resp = c.get_response(ids[i]); ref = orc.arr(oracle.orc_kcf_response(k), nb)
            assert resp.argmax() == ref.argmax()
            assert abs(resp.max() - ref.max()) <= PEAK_RTOL * abs(ref.max())
            boxes[i] = exp + (0.9,)
        c.update_batch(ids, boxes)
        for i, k in enumerate(oks):
            oracle.orc_kcf_update(k, P(orc.crop_patch(oracle, frame, boxes[i], rows, cols)), P(orc.boxes_array([boxes[i]])))
    for k in oks:
        oracle.orc_kcf_delete(k)
    c.close()


def test_kcf_batch_from_frame_vs_oracle(mot, oracle):
    """64 tracks in one launch, crops taken on device from the bound frame (td.cpp:344-384,512-582)."""
    from multiple_object_tracking_amd import synth
    scene = synth.Scene(64, 80, stream_id=7)
    c = mot.MotContext(max_tracks=64, max_dets=64)
    frames = list(scene.frames(4))
    frame0, dets0 = frames[0]
    c.frame_upload(frame0)
    ids = c.tracks_new(dets0)
    oks = []
    for d in dets0:
        b = orc.boxes_array([d])
        k = C.c_void_p(oracle.orc_kcf_new(P(b), 0))
        patch = orc.crop_patch(oracle, frame0, d, 80, 80)
        oracle.orc_kcf_update(k, P(patch), P(b))
        oks.append(k)
    boxes = [tuple(d) for d in dets0]
    for frame, _ in frames[1:]:
        c.frame_upload(frame)
        pred = c.predict_batch(ids, clamp=True)
        for i, k in enumerate(oks):
            patch = orc.crop_patch(oracle, frame, boxes[i], 80, 80)
            pb = BBox()
            oracle.orc_kcf_predict(k, P(patch), C.byref(pb))
            exp = (min(max(pb.l, 0), 1279), min(max(pb.t, 0), 719), min(max(pb.b, 0), 719), min(max(pb.r, 0), 1279), pb.type)
            assert tuple(boxes_to_np(pred[i:i + 1])[0]) == exp, f"track {i}"
            resp = c.get_response(ids[i]); ref = orc.arr(oracle.orc_kcf_response(k), 400)
            assert resp.argmax() == ref.argmax()
            assert abs(resp.max() - ref.max()) <= PEAK_RTOL * abs(ref.max())
            boxes[i] = exp + (0.9,)
        c.update_batch(ids, boxes)
        for i, k in enumerate(oks):
            patch = orc.crop_patch(oracle, frame, boxes[i], 80, 80)
            oracle.orc_kcf_update(k, P(patch), P(orc.boxes_array([boxes[i]])))
    for k in oks:
        oracle.orc_kcf_delete(k)
    c.close()


# ---------------------------------------------------------------- Kalman ----
def test_kalman_golden_and_oracle(mot, oracle):
    g = load("kalman_cases.npz")
    ntr, nst = g["z"].shape[:2]
    c = mot.MotContext(tracker_kind=mot.TRACKER_KALMAN, max_tracks=16, max_dets=16)
    ids = c.tracks_new([tuple(int(v) for v in g["box0"][tr]) + (0, 0.9) for tr in range(ntr)])
    cur = [tuple(int(v) for v in g["box0"][tr]) + (3, 0.5) for tr in range(ntr)]
    for s in range(nst):
        pred = c.predict_batch(ids, clamp=False, boxes_inout=cur)
        for tr in range(ntr):
            assert tuple(boxes_to_np(pred[tr:tr + 1])[0][:4]) == tuple(int(v) for v in g["pred"][tr, s]), f"track {tr} step {s}"
            assert pred[tr]["type"] == 3           # predict writes only l,t,r,b (kalman.cpp:112-115)
        z = [tuple(int(v) for v in g["z"][tr, s]) + (0, 0.9) for tr in range(ntr)]
        c.update_batch(ids, z)
        for tr in range(ntr):
            x, Pm = c.get_kalman_state(ids[tr])
            np.testing.assert_allclose(x, g["x"][tr, s], rtol=1e-11, atol=1e-11)      # vs reference (its BLAS sums differently)
            np.testing.assert_allclose(Pm, g["P"][tr, s], rtol=1e-11, atol=1e-8)
    # vs oracle at the stated 1e-12
    rng = np.random.default_rng(3)
    b0 = (300, 200, 279, 379, 1, 0.9)
    idk = c.tracks_new([b0])
    k = C.c_void_p(oracle.orc_kalman_new(P(orc.boxes_array([b0]))))
    for s in range(40):
        pred = c.predict_batch(idk, clamp=False, boxes_inout=[b0])
        pb = BBox(); oracle.orc_kalman_predict(k, C.byref(pb))
        assert tuple(boxes_to_np(pred)[0][:4]) == (pb.l, pb.t, pb.b, pb.r)
        z = (300 + 2 * s + int(rng.integers(-3, 4)), 200 - s + int(rng.integers(-3, 4)), 279 - s, 379 + 2 * s, 1, 0.9)
        c.update_batch(idk, [z]); oracle.orc_kalman_update(k, P(orc.boxes_array([z])))
        x, Pm = c.get_kalman_state(idk[0])
        xo = np.zeros(6); Po = np.zeros(36); oracle.orc_kalman_get_state(k, P(xo), P(Po))
        np.testing.assert_allclose(x, xo, rtol=KALMAN_RTOL, atol=1e-12)
        np.testing.assert_allclose(Pm, Po, rtol=KALMAN_RTOL, atol=1e-12 * np.abs(Po).max())
    c.close()


# ---------------------------------------------------------------- cost ------
def test_cost_matrix_bit_exact(mot, oracle):
    c = mot.MotContext(max_tracks=1024, max_dets=1024)
    # every integer dx in 0..1023 against every dy in 0..719 -> exercises the device float64 sqrt on ~0.5M distinct arguments
    trk = [(2 * i, 0, 0, 0, i % 3, 0.9) for i in range(1024)]       # centre x = i, y = 0
    det = [(0, 2 * j, 2 * j, 0, j % 3, 0.9) for j in range(720)]    # centre x = 0, y = j
    got = c.cost_matrix(trk, det)
    ref = orc.cost_matrix(oracle, trk, det)
    assert np.array_equal(got.view(np.uint64), ref.view(np.uint64))
    rng = np.random.default_rng(9)
    for nT, nD in [(16, 16), (5, 40), (40, 5), (64, 64), (300, 17)]:
        def rb(n):
            l = rng.integers(0, 1200, n); t = rng.integers(0, 640, n)
            return [(int(l[i]), int(t[i]), int(t[i]) + 79, int(l[i]) + 79, int(rng.integers(0, 3)), 0.9) for i in range(n)]
        tb, db = rb(nT), rb(nD)
        assert np.array_equal(c.cost_matrix(tb, db).view(np.uint64), orc.cost_matrix(oracle, tb, db).view(np.uint64))
    c.close()


# ---------------------------------------------------------------- Munkres ---
def _mm(rng, nr, nc, kind):
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


def test_munkres_golden_bit_exact(mot):
    c = mot.MotContext(max_tracks=1024, max_dets=1024)
    g = load("munkres_cases.npz")
    for i in range(int(g["n"])):
        nr, nc, _ = map(int, g[f"m{i}_shape"])
        a, cost = c.assignment_optimal(g[f"m{i}_d"], nr, nc)
        assert np.array_equal(a, g[f"m{i}_a"]), f"golden matrix {i} ({nr}x{nc})"
        assert cost == float(g[f"m{i}_c"]), f"golden matrix {i} cost"
    for i in range(int(g["nbig"])):
        nn, kind, seed = map(int, g[f"big{i}_spec"])
        d = _mm(np.random.default_rng(seed), nn, nn, kind)
        a, cost = c.assignment_optimal(d, nn, nn)
        assert np.array_equal(a, g[f"big{i}_a"]), f"big {i}"
        assert cost == float(g[f"big{i}_c"])
    c.close()


def test_munkres_random_vs_oracle(mot, oracle):
    c = mot.MotContext(max_tracks=1024, max_dets=1024)
    rng = np.random.default_rng(77)
    for trial in range(160):
        nr = int(rng.integers(1, 130)); nc = int(rng.integers(1, 130))
        if trial % 4 == 0:
            nc = nr
        d = _mm(rng, nr, nc, trial % 6)
        a, cost = c.assignment_optimal(d, nr, nc)
        ra, rc = orc.assignment_optimal(oracle, d, nr, nc)
        assert np.array_equal(a, ra), f"trial {trial} {nr}x{nc} kind {trial % 6}"
        assert cost == rc
    # empty / degenerate
    a, cost = c.assignment_optimal(np.zeros(0), 0, 5)
    assert len(a) == 0 and cost == 0.0
    c.close()


@pytest.mark.parametrize("n", [64, 256, 1024])
def test_assign_tracking_costs_vs_oracle(mot, oracle, n):
    """td.cpp cost matrix + Munkres on device for tracking-like box sets (rows = smaller side)."""
    c = mot.MotContext(max_tracks=1024, max_dets=1024)
    rng = np.random.default_rng(n)
    for nT, nD in [(n, n), (n, max(n - 7, 1)), (max(n - 5, 1), n)]:
        cx = rng.integers(0, 1200, size=max(nT, nD)); cy = rng.integers(0, 640, size=max(nT, nD))
        trk = [(int(cx[i] + rng.integers(-3, 4)), int(cy[i] + rng.integers(-3, 4)), int(cy[i]) + 79, int(cx[i]) + 79, i % 3, 0.9) for i in range(nT)]
        perm = rng.permutation(max(nT, nD))[:nD]
        det = [(int(cx[i] + rng.integers(-2, 3)), int(cy[i] + rng.integers(-2, 3)), int(cy[i]) + 79, int(cx[i]) + 79, int(i % 3), 0.9) for i in perm]
        at, ad, cost = c.assign(trk, det)
        d = orc.cost_matrix(oracle, trk, det)
        if nT < nD:
            ra, rc = orc.assignment_optimal(oracle, d, nT, nD)
            assert np.array_equal(at, ra)
        else:
            ra, rc = orc.assignment_optimal(oracle, d, nD, nT)
            exp = np.full(nT, -1, np.int32)
            for j in range(nD):
                exp[ra[j]] = j
            assert np.array_equal(at, exp)
            assert np.array_equal(ad, ra)
        assert cost == rc
    c.close()


def test_assign_row_scan_fallbacks_vs_oracle(mot, oracle):
    """lap_rowscan_kernel keeps the 8 nearest SAME-CLASS columns of a row through packed integer keys (squared distance << 10 | column,
    LDS-staged centroids) and must fall back to its general float64 form whenever that is not the whole truth: a class with fewer than 8
    members (the row's 8 smallest costs then include cross-class entries), boxes outside the +-1400 px range the packed centroids cover,
    and centroid distances beyond 2048 px.  Every variant must reproduce the oracle's assignment and cost."""
    c = mot.MotContext(max_tracks=1024, max_dets=1024)
    rng = np.random.default_rng(77)
    n = 160
    def boxes(off_frame=False, rare=False, far=False):
        cx = rng.integers(0, 1200, size=n); cy = rng.integers(0, 640, size=n)
        ty = np.array([i % 3 for i in range(n)])
        if rare: ty = np.where(np.arange(n) < 5, 1, 0)                   # class 1 has five members only
        if off_frame: cx[:6] += 2000                                     # beyond box_small's range
        if far: cx[:4] -= 1300; cx[4:8] += 900                           # pairs farther apart than 2048 px exist
        trk = [(int(cx[i] + rng.integers(-3, 4)), int(cy[i] + rng.integers(-3, 4)), int(cy[i]) + 79, int(cx[i]) + 79, int(ty[i]), 0.9) for i in range(n)]
        perm = rng.permutation(n)
        det = [(int(cx[i] + rng.integers(-2, 3)), int(cy[i] + rng.integers(-2, 3)), int(cy[i]) + 79, int(cx[i]) + 79, int(ty[i]), 0.9) for i in perm]
        return trk, det
    for kw in ({}, {"rare": True}, {"off_frame": True}, {"far": True}, {"rare": True, "far": True}):
        trk, det = boxes(**kw)
        at, ad, cost = c.assign(trk, det)
        ra, rc = orc.assignment_optimal(oracle, orc.cost_matrix(oracle, trk, det), n, n)
        assert np.array_equal(ad, ra) and cost == rc, kw
    c.close()


# ---------------------------------------------------------------- frame loop
def _scene(spec):
    from multiple_object_tracking_amd import synth
    n, size, sid, miss, fp, nframes = map(int, spec[:6])
    ds = (int(spec[6]), int(spec[7])) if len(spec) > 6 else None
    return synth.Scene(n, size, stream_id=sid, det_sizes=ds, miss_pct=miss, fp_pct=fp), nframes


@pytest.mark.parametrize("name,kind", [("frameloop_kalman.npz", 1), ("frameloop_kcf.npz", 0), ("frameloop_kcf_multiscale.npz", 0)])
def test_step_frame_golden_trace(mot, name, kind):
    """mot_step_frame (td.cpp:344-644 on device) against the trace recorded from the reference's functions.
    frameloop_kalman is BASELINE config 1 (16 Kalman tracks + 16 detections)."""
    g = load(name)
    scene, nframes = _scene(g["spec"])
    c = mot.MotContext(tracker_kind=kind, max_tracks=256, max_dets=128)
    for f, (frame, dets) in enumerate(scene.frames(nframes)):
        if kind == 0:
            c.frame_upload(frame)
        out = c.step_frame(dets)
        assert np.array_equal(boxes_to_np(out["predicted"]), g[f"f{f}_pred"]), f"frame {f} predicted boxes"
        assert np.array_equal(out["assigned"], g[f"f{f}_assigned"]), f"frame {f} assignment"
        assert np.array_equal(boxes_to_np(out["live"]), g[f"f{f}_live"]), f"frame {f} live tracks"
        assert np.array_equal(out["tids"].astype(np.int32), g[f"f{f}_tids"])
    c.close()


def test_step_frame_64_tracks_vs_oracle(mot, oracle):
    """BASELINE config 2 shape (64 KCF tracks, 80x80) for a few frames against the oracle frame loop."""
    from multiple_object_tracking_amd import synth
    scene = synth.Scene(64, 80, stream_id=11, miss_pct=3, fp_pct=2)
    c = mot.MotContext(max_tracks=128, max_dets=128)
    m = orc.OracleMot(oracle, 0, 0, 128)
    for f, (frame, dets) in enumerate(scene.frames(5)):
        c.frame_upload(frame)
        out = c.step_frame(dets)
        ref = m.step(frame, dets)
        assert np.array_equal(boxes_to_np(out["predicted"]), boxes_to_np(ref["predicted"])), f"frame {f}"
        assert np.array_equal(out["assigned"], ref["assigned"]), f"frame {f}"
        assert np.array_equal(boxes_to_np(out["live"]), boxes_to_np(ref["live"])), f"frame {f}"
        assert np.array_equal(out["tids"], ref["tids"])
    m.close(); c.close()


# ---------------------------------------------------------------- sharding --
def test_two_rank_shard_matches_single(mot):
    """tracks sharded tid % 2 over two contexts (one GPU, one process): after exchanging the predicted
    boxes both ranks must reproduce the unsharded run exactly."""
    from multiple_object_tracking_amd import synth
    hip = C.CDLL("libamdhip64.so")
    scene = synth.Scene(24, 80, stream_id=5, miss_pct=5, fp_pct=3)
    single = mot.MotContext(max_tracks=64, max_dets=64)
    ranks = [mot.MotContext(max_tracks=64, max_dets=64, rank=r, world=2) for r in range(2)]
    for f, (frame, dets) in enumerate(scene.frames(6)):
        for c in [single] + ranks:
            c.frame_upload(frame)
        ref = single.step_frame(dets)
        segs = [c.step_begin() for c in ranks]
        spr = segs[0][1]
        # "all-gather": rank r's segment -> slot r of every rank's gather buffer (same layout as ncclAllGather)
        for dst_r, c in enumerate(ranks):
            base = segs[dst_r][0] - dst_r * spr * 24
            for src_r in range(2):
                if src_r != dst_r:
                    assert hip.hipMemcpy(C.c_void_p(base + src_r * spr * 24), C.c_void_p(segs[src_r][0]), spr * 24, 3) == 0
        for r, c in enumerate(ranks):
            base = segs[r][0] - r * spr * 24
            out = c.step_finish(base, dets)
            assert np.array_equal(boxes_to_np(out["predicted"]), boxes_to_np(ref["predicted"])), f"frame {f} rank {r}"
            assert np.array_equal(out["assigned"], ref["assigned"])
            assert np.array_equal(boxes_to_np(out["live"]), boxes_to_np(ref["live"]))
            assert np.array_equal(out["tids"], ref["tids"])
    for c in [single] + ranks:
        c.close()


# ---------------------------------------------------------------- drop-in ---
def test_dropin_per_object_interface(mot):
    """the reference's five symbols (td.cpp:229-234), resolved by their mangled names."""
    g = load("kcf_seq_80.npz")
    lib = C.CDLL(mot.DROPIN_KCF_PATH)
    new = getattr(lib, "_Z11tracker_newP11_bbox_pos_s"); new.restype = C.c_void_p; new.argtypes = [C.c_void_p]
    pred = getattr(lib, "_Z15tracker_predictPvPfP11_bbox_pos_s"); pred.argtypes = [C.c_void_p] * 3; pred.restype = None
    upd = getattr(lib, "_Z14tracker_updatePvPfP11_bbox_pos_s"); upd.argtypes = [C.c_void_p] * 3; upd.restype = None
    dele = getattr(lib, "_Z14tracker_deletePv"); dele.argtypes = [C.c_void_p]; dele.restype = None
    asg = getattr(lib, "_Z17assignmentoptimalPiPdS0_ii"); asg.restype = None
    l, t, b, r, ty = map(int, g["box0"])
    b0 = mot.BBox(l, t, b, r, ty, 0.9)
    h = C.c_void_p(new(C.byref(b0)))
    p = np.ascontiguousarray(g["p_init"]); upd(h, P(p), C.byref(b0))
    for s in range(int(g["steps"])):
        p = np.ascontiguousarray(g[f"s{s}_patch"]); pb = mot.BBox()
        pred(h, P(p), C.byref(pb))
        assert (pb.l, pb.t, pb.b, pb.r, pb.type) == tuple(int(v) for v in g[f"s{s}_pred"])
        ul, ut, ub, ur, uty = map(int, g[f"s{s}_ubox"]); nb = mot.BBox(ul, ut, ub, ur, uty, 0.9)
        upd(h, P(p), C.byref(nb))
    dele(h)
    gm = load("munkres_cases.npz")
    for i in (5, 9, 13):
        nr, nc, _ = map(int, gm[f"m{i}_shape"])
        a = np.zeros(nr, np.int32); cost = C.c_double(0); d = np.ascontiguousarray(gm[f"m{i}_d"])
        asg(P(a), C.byref(cost), P(d), nr, nc)
        assert np.array_equal(a, gm[f"m{i}_a"]) and cost.value == float(gm[f"m{i}_c"])


def test_dropin_c_helpers_on_device(mot, oracle):
    """rgb2Gray / bilinearInterpolationGray / drawRect (td.cpp:235-261, C linkage) as exported by the drop-in libraries: device kernels
    behind the reference's host-pointer signatures.  Crop + resize composed as td.cpp:346-364 does must reproduce the reference-generated
    crop fixtures bit for bit; three nested outlines per track as td.cpp:701-731 draws them must reproduce the reference-generated overlay
    frames byte for byte."""
    import sys
    sys.path.insert(0, os.path.join(orc.ROOT, "tests", "golden"))
    from make_overlay_golden import hashcolor
    for path in (mot.DROPIN_KCF_PATH, mot.DROPIN_KALMAN_PATH):
        lib = C.CDLL(path)
        lib.rgb2Gray.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int32] * 4; lib.rgb2Gray.restype = None
        lib.bilinearInterpolationGray.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4; lib.bilinearInterpolationGray.restype = None
        lib.drawRect.argtypes = [C.c_void_p] + [C.c_int32] * 4 + [C.c_uint32]; lib.drawRect.restype = None
        g = load("crop_cases.npz")
        frame = np.random.default_rng(int(g["seed"])).integers(0, 256, size=(720, 1280, 3), dtype=np.uint8)
        for i in range(int(g["n"])):
            l, t, r, b, rows, cols = map(int, g[f"c{i}_box"])
            scratch = np.zeros((b - t + 1) * (r - l + 1), np.float32); patch = np.zeros(rows * cols, np.float32)
            lib.rgb2Gray(P(scratch), P(frame), l, t, r, b)
            lib.bilinearInterpolationGray(P(patch), P(scratch), b - t + 1, r - l + 1, rows, cols)
            assert np.array_equal(patch.view(np.uint32), g[f"c{i}_patch"].view(np.uint32)), f"crop case {i}"
        go = np.load(os.path.join(orc.ROOT, "tests", "golden", "overlay_cases.npz"))
        cm = go["colormap"]
        for k in range(int(go["n"])):
            fr = np.zeros(720 * 1280 * 3, np.uint8)
            for bb, tid in zip(go[f"boxes_{k}"], go[f"tids_{k}"]):
                color = int(cm[hashcolor((int(tid) + 1) & 0xFFFFFFFF) & 255])
                for d in range(3):
                    lib.drawRect(P(fr), int(bb["l"]) + d, int(bb["t"]) + d, int(bb["r"]) - d, int(bb["b"]) - d, color)
            exp = np.zeros(720 * 1280 * 3, np.uint8); exp[go[f"idx_{k}"]] = go[f"val_{k}"]
            assert np.array_equal(fr, exp), f"overlay case {k}"


_VARIANT_CODE = r'''
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import mot_amd, orc
g = np.load(os.path.join("tests", "golden", "munkres_cases.npz"))
c = mot_amd.MotContext(max_tracks=1024, max_dets=1024)
for i in range(int(g["n"])):
    nr, nc, _ = map(int, g[f"m{i}_shape"])
    a, cost = c.assignment_optimal(g[f"m{i}_d"], nr, nc)
    assert np.array_equal(a, g[f"m{i}_a"]) and cost == float(g[f"m{i}_c"]), i
lib = orc.load_oracle(); rng = np.random.default_rng(5)
for n in (64, 300, 1024):
    cx = rng.integers(0, 1200, n); cy = rng.integers(0, 640, n)
    trk = [(int(cx[i] + rng.integers(-3, 4)), int(cy[i] + rng.integers(-3, 4)), int(cy[i]) + 79, int(cx[i]) + 79, i % 3, 0.9) for i in range(n)]
    det = [(int(cx[i] + rng.integers(-2, 3)), int(cy[i] + rng.integers(-2, 3)), int(cy[i]) + 79, int(cx[i]) + 79, int(i % 3), 0.9) for i in rng.permutation(n)]
    at, ad, cost = c.assign(trk, det)
    ra, rc = orc.assignment_optimal(lib, orc.cost_matrix(lib, trk, det), n, n)
    assert np.array_equal(ad, ra) and cost == rc, n
for n in (257, 520, 1024):            # dense uniform costs: hundreds of step-5 passes with hundreds of uncovered columns
    d = rng.uniform(0, 1, size=n * n)
    a, cost = c.assignment_optimal(d, n, n)
    ra, rc = orc.assignment_optimal(lib, d, n, n)
    assert np.array_equal(a, ra) and cost == rc, ("uniform", n)
    assert c.assoc_stats()[15] == 0, ("helper protocol timed out", n)
d = rng.uniform(0, 1, size=700 * 400)  # rectangular, both orientations
for nr, nc in ((700, 400), (400, 700)):
    a, cost = c.assignment_optimal(d, nr, nc)
    ra, rc = orc.assignment_optimal(lib, d, nr, nc)
    assert np.array_equal(a, ra) and cost == rc, ("rect", nr, nc)
# random matrix families incl. tie-heavy ones (the fast path must hand every tie to the order-exact emulation)
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from test_lap_model import mm
for trial in range(120):
    nr = int(rng.integers(1, 140)); nc = int(rng.integers(nr, 150))
    d = mm(rng, nr, nc, trial % 9)
    a, cost = c.assignment_optimal(d, nr, nc)
    ra, rc = orc.assignment_optimal(lib, d, nr, nc)
    assert np.array_equal(a, ra) and cost == rc, ("family", trial % 9, nr, nc)
print("VARIANT_OK", c.assoc_stats()[:3].tolist(), "LAP", c.lap_stats()[16:28].tolist())
'''


def _run_variant(env_extra):
    import subprocess, sys
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, "-c", _VARIANT_CODE], cwd=orc.ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert "VARIANT_OK" in out.stdout, out.stdout + out.stderr
    return eval(out.stdout.split("LAP")[-1].strip())


@pytest.mark.parametrize("helpers", ["0", "1", "2"])
def test_munkres_helper_workgroups_subprocess(helpers):
    """step-5 helper workgroups (munkres_kernel<true>, 1 + 16 workgroups, cross-CU control block) forced on for every
    problem above 256 lines ("1") and forced off ("0"): identical assignments and cost either way.  "2": on, and every helper reports
    a zero among its (covered row, covered column) entries, so the controller merges the per-row COVBITS granules instead of clearing
    those bits itself -- the path a pathological rounding case would take."""
    _run_variant({"MOT_MUNKRES_HELPERS": helpers})


@pytest.mark.parametrize("mode", ["all_sizes", "all_sizes_one_event_loop", "all_sizes_full_reset", "off"])
def test_lap_fast_path_subprocess(mode):
    """assignment fast path (lap_kernels.hip: exact sparse solver + uniqueness certificate) forced on for EVERY problem size
    (MOT_LAP_MIN=1) and switched off (MOT_LAP_FAST=0): bit-identical assignments and cost either way, and with it on both
    outcomes must occur -- certified launches (emulation skipped) and tied optima (order-exact emulation ran)."""
    # all_sizes_one_event_loop: the sparse emulation's event loop without batching (MOT_MK_BATCH=0): one step-3 event per iteration
    # all_sizes_full_reset: the sparse emulation with the reference's full reset after every augmentation (MOT_MK_LAZY=0) instead of the lazy one
    env = {"off": {"MOT_LAP_FAST": "0"}, "all_sizes": {"MOT_LAP_MIN": "1"}, "all_sizes_one_event_loop": {"MOT_LAP_MIN": "1", "MOT_MK_BATCH": "0"},
           "all_sizes_full_reset": {"MOT_LAP_MIN": "1", "MOT_MK_LAZY": "0"}}[mode]
    cum = _run_variant(env)
    if mode == "off":
        assert sum(cum) == 0, cum
    else:
        assert cum[0] > 20 and cum[4] > 20, cum                     # certified launches and tied optima both occurred
        assert cum[2] <= 5, cum                                     # dense random matrices can defeat the sparse solver's duals (then the emulation decides); rare
        assert cum[8] > 20 and cum[9] > 20, cum                     # uncertified problems: sparse emulation accepted / refused (dense emulation ran)


def test_lap_fast_path_counters(mot, oracle):
    """default configuration: a crowded 1024 x 1024 tracking problem with a unique optimum is certified (no emulation:
    step counters stay at their 'skipped' mark), the same problem with two tracks on one centroid is a tie and goes through
    the sparse order-exact emulation, a dense random matrix through the dense one; all equal the oracle"""
    c = mot.MotContext(max_tracks=1024, max_dets=1024)
    rng = np.random.default_rng(12)
    n = 1024
    while True:
        cx = rng.permutation(1200 * 640)[:n]
        px, py = cx % 1200, cx // 1200                                # distinct centroids
        trk = [(int(px[i]) - 40, int(py[i]) - 40, int(py[i]) + 39, int(px[i]) + 39, i % 3, 0.9) for i in range(n)]
        dobj = [(int(px[i] + rng.integers(-3, 4)) - 40, int(py[i] + rng.integers(-3, 4)) - 40, int(py[i]) + 39, int(px[i]) + 39, int(i % 3), 0.9) for i in range(n)]
        perm = rng.permutation(n)
        det = [dobj[i] for i in perm]
        at, ad, cost = c.assign(trk, det)
        ra, rc = orc.assignment_optimal(oracle, orc.cost_matrix(oracle, trk, det), n, n)
        assert np.array_equal(ad, ra) and cost == rc
        if c.lap_stats()[0] == 0:
            break                                                     # (a random scene may contain a symmetric tie: draw again)
    assert c.assoc_stats()[0] == -1 and c.assoc_stats()[1] == 0       # certified: no step 4 / step 5 ran
    # object 7 moves onto object 3: two tracks on ONE centroid and class (identical cost columns), their two detections next to
    # each other -> two optimal assignments; which one the reference returns is decided by its scan order
    trk[7] = trk[3]
    d3 = dobj[3]
    dobj[7] = (d3[0] + 1, d3[1], d3[2], d3[3] + 1, d3[4], 0.9)
    det = [dobj[i] for i in perm]
    at, ad, cost = c.assign(trk, det)
    ra, rc = orc.assignment_optimal(oracle, orc.cost_matrix(oracle, trk, det), n, n)
    assert np.array_equal(ad, ra) and cost == rc
    st = c.lap_stats()
    assert st[0] == 4 and st[6] >= 2, st[:8]                           # the certificate refused: tied optima ...
    assert st[15] == 1 and st[8] == 0 and st[9] > 0, st[8:16]          # ... and the sparse order-exact emulation decided the tie
    # a dense random matrix defeats the candidate lists: the sparse run is refused after the fact, the dense emulation decides
    d = rng.uniform(0, 1, size=300 * 300)
    a, cost = c.assignment_optimal(d, 300, 300)
    ra, rc = orc.assignment_optimal(oracle, d, 300, 300)
    assert np.array_equal(a, ra) and cost == rc
    assert c.lap_stats()[15] == 2 and c.assoc_stats()[0] >= 0
    c.close()


def test_assign_stress_short():
    """tools/assign_stress.py for 20 s: random crowded scenes with duplicated centroids (tied optima), rectangular shapes, 97..1024
    lines -- every assignment and cost bit-equal to the oracle, whichever tier decided (a 240 s run: 22,933 problems, decided by
    certificate 11,205 / sparse emulation 11,692 / dense emulation 36)"""
    import subprocess, sys
    out = subprocess.run([sys.executable, os.path.join(orc.ROOT, "tools", "assign_stress.py"), "20", "3"], cwd=orc.ROOT, capture_output=True, text=True, timeout=600)
    assert "assign_stress OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_step_frame_chain_entry_point(mot, oracle):
    """mot_step_frame_chain: the frame loop fed with the reference's bbox_chain_t (cnntype.h:43-47, td.cpp:330-333) -- same
    results as the oracle loop; nbox outside 0..128 is refused"""
    from multiple_object_tracking_amd import synth

    class Chain(C.Structure):
        _fields_ = [("nbox", C.c_int), ("bbox", mot.BBox * 128)]
    assert C.sizeof(Chain) == 3076
    lib = mot.load_library()
    scene = synth.Scene(24, 80, stream_id=77, miss_pct=5, fp_pct=5)
    c = mot.MotContext(max_tracks=128, max_dets=128)
    m = orc.OracleMot(oracle, 0, 0, 128)
    for f, (frame, dets) in enumerate(scene.frames(5)):
        c.frame_upload(frame)
        ch = Chain(); ch.nbox = len(dets)
        for i, d in enumerate(dets):
            ch.bbox[i] = mot.BBox(int(d[0]), int(d[1]), int(d[2]), int(d[3]), int(d[4]), 0.9)
        cap = 129
        pred = np.zeros(cap, mot.BBOX_DTYPE); at = np.zeros(cap, np.int32); live = np.zeros(cap, mot.BBOX_DTYPE); tids = np.zeros(cap, np.uint32)
        nb, nl = C.c_int(0), C.c_int(0)
        rc = lib.mot_step_frame_chain(c._h, C.byref(ch), P(pred), P(at), C.byref(nb), P(live), P(tids), C.byref(nl))
        assert rc == 0, lib.mot_last_error()
        ref = m.step(frame, dets)
        assert np.array_equal(tids[:nl.value], ref["tids"]) and np.array_equal(at[:nb.value], ref["assigned"])
        for k in ("l", "t", "b", "r", "type"):
            assert np.array_equal(live[:nl.value][k], ref["live"][k])
    bad = Chain(); bad.nbox = 129
    assert lib.mot_step_frame_chain(c._h, C.byref(bad), None, None, None, None, None, None) == -1
    m.close(); c.close()


