"""Overlay step (top/td.cpp:647-733 + top/drawlib.c:97-151): the oracle restatement against vectors produced by the
reference's own drawRect (tests/golden/make_overlay_golden.py), and the device kernels against both -- every byte of the frame."""
import ctypes as C
import os

import numpy as np
import pytest

import orc

FIX = os.path.join(orc.ROOT, "tests", "golden", "overlay_cases.npz")
NB = 720 * 1280 * 3


def _expected(g, k):
    f = np.zeros(NB, np.uint8)
    f[g[f"idx_{k}"]] = g[f"val_{k}"]
    return f


def _oracle_overlay(lib, frame, boxes, tids):
    lib.orc_overlay(orc.P(frame), orc.P(boxes), orc.P(np.ascontiguousarray(tids, np.uint32)), len(boxes))


def test_oracle_overlay_matches_reference_vectors(oracle):
    g = np.load(FIX)
    oracle.orc_colormap.restype = C.POINTER(C.c_uint32)
    cm = np.ctypeslib.as_array(oracle.orc_colormap(), shape=(256,))
    assert np.array_equal(cm, g["colormap"])                            # td.cpp:655-697, incl. its two non-xterm greys
    for k in range(int(g["n"])):
        frame = np.zeros(NB, np.uint8)
        _oracle_overlay(oracle, frame, np.ascontiguousarray(g[f"boxes_{k}"]), g[f"tids_{k}"])
        assert np.array_equal(frame, _expected(g, k)), f"case {k}"


def test_overlay_colour_follows_the_spawn_sequence(oracle):
    """td.cpp:619-620: `tid = tracker_id++; color = hashcolor(tracker_id) & 255` -- the colour is hashed from the id AFTER the
    increment, so the track with tid 0 is painted in colormap[hashcolor(1) & 255] (round-3 advisor finding: tid was hashed)."""
    from golden.make_overlay_golden import hashcolor
    g = np.load(FIX)
    cm = g["colormap"]
    for tid in (0, 1, 7, 255, 2 ** 32 - 1):
        frame = np.zeros(NB, np.uint8)
        boxes = orc.boxes_array([(100, 50, 90, 160, 0, 0.9)])
        _oracle_overlay(oracle, frame, boxes, np.array([tid], np.uint32))
        px = frame.reshape(720, 1280, 3)[50, 130]                       # a pixel of the top edge; drawRect stores R, G, B at byte offsets 0, 1, 2
        want = int(cm[hashcolor((tid + 1) & 0xFFFFFFFF) & 255])
        assert (int(px[0]) << 16 | int(px[1]) << 8 | int(px[2])) == want, tid
    assert cm[hashcolor(1) & 255] != cm[hashcolor(0) & 255]              # (so the test can tell the two apart)


@pytest.mark.gpu
def test_device_overlay_matches_reference_vectors_and_oracle(mot, oracle):
    import torch
    g = np.load(FIX)
    c = mot.MotContext(max_tracks=64, max_dets=64)
    for k in range(int(g["n"])):
        fd = torch.zeros(NB, dtype=torch.uint8, device="cuda")
        for rep in range(2):                                            # twice: the epoch-tagged stamps must not leak between calls
            fd.zero_()
            c.overlay_draw(fd.data_ptr(), g[f"boxes_{k}"], g[f"tids_{k}"]); c.sync()
            assert np.array_equal(fd.cpu().numpy(), _expected(g, k)), f"case {k} rep {rep}"
    # random scenes on a non-zero frame: 1000 tracks, boxes partly outside the frame (skipped pixels), equal to the oracle byte for byte
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, NB).astype(np.uint8)
    n = 1000
    l = rng.integers(-30, 1270, n); t = rng.integers(-30, 710, n); w = rng.integers(1, 120, n); h = rng.integers(1, 120, n)
    boxes = mot.boxes_array([(int(l[i]), int(t[i]), int(t[i] + h[i]), int(l[i] + w[i]), 0, 0.9) for i in range(n)])
    tids = rng.integers(0, 2 ** 32, n).astype(np.uint32)
    exp = base.copy(); _oracle_overlay(oracle, exp, boxes, tids)
    fd = torch.from_numpy(base.copy()).cuda()
    c.overlay_draw(fd.data_ptr(), boxes, tids); c.sync()
    assert np.array_equal(fd.cpu().numpy(), exp)
    c.close()


@pytest.mark.gpu
def test_device_overlay_live_list(mot, oracle):
    """mot_overlay_live: the device-resident loop's own live list (boxes, track ids, count all in HBM) drawn without a copy"""
    import torch
    from multiple_object_tracking_amd import synth
    scene = synth.Scene(200, 80, stream_id=31, miss_pct=5, fp_pct=5)
    c = mot.MotContext(max_tracks=256, max_dets=256)
    for f, (frame, dets) in enumerate(scene.frames(4)):
        fd = torch.from_numpy(frame).cuda()
        da = mot.boxes_array(dets); dd = torch.from_numpy(da.view(np.uint8)).cuda()
        c.step_frame_device(fd.data_ptr(), dd.data_ptr(), len(dets))
        boxes, tids, _ = c.live_tracks()
        out = torch.from_numpy(frame.copy()).cuda()
        c.overlay_live(out.data_ptr()); c.sync()
        exp = frame.copy().reshape(-1); _oracle_overlay(oracle, exp, np.ascontiguousarray(boxes), tids)
        assert np.array_equal(out.cpu().numpy().reshape(-1), exp), f"frame {f}"
    c.close()
