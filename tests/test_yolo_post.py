"""Detector post-processing (detectors/yolo3.cpp:141-356 + 490-547): device kernels against the oracle restatement.
PARITY UNPINNED: yolo3.cpp needs <Windows.h> and TensorFlow and cannot be compiled here, and the reference has no vectors for it;
the restatement follows the source by reading (incl. its exchange sort and the never-reset `is_suppressed` vector of do_nms)."""
import ctypes as C

import numpy as np
import pytest

import orc

ANCHORS = [10, 13, 16, 30, 33, 23, 30, 61, 62, 45, 59, 119, 116, 90, 156, 198, 373, 326]


def _heads(rng, th, tw, nc, npos):
    heads = []
    for s in range(3):
        gh, gw = (th // 32) << s, (tw // 32) << s
        h = rng.normal(0.0, 1.0, (gh, gw, 3, 5 + nc)).astype(np.float32)
        h[..., 4] = -9.0                                                # background: objectness ~ 1e-4
        h[..., 2:4] = rng.normal(0.0, 0.4, (gh, gw, 3, 2))
        for _ in range(npos):                                           # objects, clustered so that NMS has work (overlapping boxes, equal scores)
            i, j, b = int(rng.integers(0, gh)), int(rng.integers(0, gw)), int(rng.integers(0, 3))
            h[i, j, b, 4] = rng.choice([2.0, 3.0, 3.0, 5.0])            # repeated values -> tied objectness
            h[i, j, b, 5:] = -6.0; h[i, j, b, 5 + int(rng.integers(0, nc))] = rng.choice([3.0, 4.0])
            if j + 1 < gw:
                h[i, j + 1, b] = h[i, j, b]                             # a neighbour cell proposing nearly the same box
        heads.append(np.ascontiguousarray(h.reshape(-1)))
    return heads


def _oracle(oracle, heads, th, tw, nc, ih, iw, ot, nt, cap=1024):
    class Opt(C.Structure):
        _fields_ = [("obj_thresh", C.c_float), ("nms_thresh", C.c_float), ("anchors", C.c_int * 18)]
    o = Opt(); o.obj_thresh, o.nms_thresh = ot, nt
    for i, a in enumerate(ANCHORS):
        o.anchors[i] = a
    out = np.zeros(cap, orc.BBOX_DTYPE)
    oracle.orc_yolo_postprocess.restype = C.c_int
    n = oracle.orc_yolo_postprocess(orc.P(heads[0]), orc.P(heads[1]), orc.P(heads[2]), th, tw, nc, ih, iw, C.byref(o), orc.P(out), cap)
    return out[:n]


def test_oracle_yolo_postprocess_smoke(oracle):
    rng = np.random.default_rng(3)
    heads = _heads(rng, 416, 416, 3, 12)
    out = _oracle(oracle, heads, 416, 416, 3, 720, 1280, 0.5, 0.45)
    assert 5 <= len(out) <= 120
    assert (out["l"] <= out["r"]).all() and (out["t"] <= out["b"]).all() and (out["r"] < 1280).all() and (out["b"] < 720).all()
    assert (np.diff(out["type"]) >= 0).all()                            # classes ascending, as do_nms emits them


@pytest.mark.gpu
@pytest.mark.parametrize("th,tw,nc,npos,seed", [(416, 416, 3, 12, 3), (320, 608, 80, 25, 4), (416, 416, 1, 40, 5)])
def test_device_yolo_postprocess_vs_oracle(mot, oracle, th, tw, nc, npos, seed):
    import torch
    rng = np.random.default_rng(seed)
    heads = _heads(rng, th, tw, nc, npos)
    exp = _oracle(oracle, heads, th, tw, nc, 720, 1280, 0.5, 0.45)
    c = mot.MotContext(max_tracks=64, max_dets=64)
    hd = [torch.from_numpy(h).cuda() for h in heads]
    dets = torch.zeros(1024 * 24, dtype=torch.uint8, device="cuda"); nd = torch.zeros(1, dtype=torch.int32, device="cuda")
    for rep in range(2):                                                # state (candidate counter, flags) must be re-armed between calls
        c.yolo_postprocess([h.data_ptr() for h in hd], th, tw, nc, 720, 1280, 0.5, 0.45, ANCHORS, dets.data_ptr(), 1024, nd.data_ptr())
        c.sync()
        n = int(nd.cpu()[0])
        got = dets.cpu().numpy().view(mot.BBOX_DTYPE)[:n]
        assert n == len(exp) and n > 0, (n, len(exp))
        for k in ("l", "t", "b", "r", "type"):
            assert np.array_equal(got[k], exp[k]), k
        assert np.allclose(got["score"], exp["score"], rtol=2e-6, atol=0)   # expf of two different maths libraries
    chain = c.yolo_postprocess([h.data_ptr() for h in hd], th, tw, nc, 720, 1280, 0.5, 0.45, ANCHORS, dets.data_ptr(), 1024, nd.data_ptr(), want_chain=True)
    assert len(chain) == min(len(exp), 128) and np.array_equal(chain["l"], exp["l"][:128])
    c.close()


@pytest.mark.gpu
def test_device_yolo_candidate_overflow_is_reported(mot):
    """more candidates above obj_thresh than the workspace holds (4096; the reference's vector is unbounded, yolo3.cpp:176-201): the call's
    output is not the reference's, and that must not pass silently -- mot_yolo_status() / the host-chain form return MOT_ERR_CAPACITY
    (round-3 advisor finding)."""
    import torch
    rng = np.random.default_rng(9)
    th = tw = 416; nc = 80
    heads = []
    for s in range(3):
        gh, gw = (th // 32) << s, (tw // 32) << s
        heads.append(np.ascontiguousarray(rng.normal(3.0, 0.5, (gh, gw, 3, 5 + nc)).astype(np.float32).reshape(-1)))   # everything is an object of every class
    c = mot.MotContext(max_tracks=64, max_dets=64)
    assert c.yolo_status() == 0
    hd = [torch.from_numpy(h).cuda() for h in heads]
    dets = torch.zeros(1024 * 24, dtype=torch.uint8, device="cuda"); nd = torch.zeros(1, dtype=torch.int32, device="cuda")
    c.yolo_postprocess([h.data_ptr() for h in hd], th, tw, nc, 720, 1280, 0.5, 0.45, ANCHORS, dets.data_ptr(), 1024, nd.data_ptr())
    with pytest.raises(mot.MotError, match="dropped"):
        c.yolo_status()
    with pytest.raises(mot.MotError, match="did not fit"):
        c.yolo_postprocess([h.data_ptr() for h in hd], th, tw, nc, 720, 1280, 0.5, 0.45, ANCHORS, dets.data_ptr(), 1024, nd.data_ptr(), want_chain=True)
    c.close()
