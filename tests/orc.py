"""ctypes access to the oracle (oracle/libmot_oracle.so) and, when present, to the
reference-built libraries in oracle/_ref.  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SRC_DIR = os.path.join(ROOT, "oracle")
# MOT_ORACLE_DIR: load the checker libraries from another build of the same sources -- oracle/_asan (make -C oracle asan: AddressSanitizer +
# UBSan, tests/test_sanitizers.py); such a directory is used as built, never rebuilt from here
ORACLE_DIR = os.environ.get("MOT_ORACLE_DIR") or ORACLE_SRC_DIR
REF_DIR = os.path.join(ORACLE_SRC_DIR, "_ref")


class BBox(C.Structure):
    _fields_ = [("l", C.c_int), ("t", C.c_int), ("b", C.c_int), ("r", C.c_int), ("type", C.c_int), ("score", C.c_float)]


BBOX_DTYPE = np.dtype([("l", "<i4"), ("t", "<i4"), ("b", "<i4"), ("r", "<i4"), ("type", "<i4"), ("score", "<f4")])
FP = C.POINTER(C.c_float)


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def build_oracle():
    so = os.path.join(ORACLE_DIR, "libmot_oracle.so")
    src = os.path.join(ORACLE_SRC_DIR, "mot_oracle.c")
    if ORACLE_DIR == ORACLE_SRC_DIR and (not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)
    return so


_orc = None


def load_oracle():
    global _orc
    if _orc is not None:
        return _orc
    lib = C.CDLL(build_oracle())
    for f in ["response", "alpha", "xm", "xf", "yf", "labels", "coswin", "features"]:
        getattr(lib, "orc_kcf_" + f).restype = FP
        getattr(lib, "orc_kcf_" + f).argtypes = [C.c_void_p]
    for f in ["rows", "cols", "frows", "fcols"]:
        getattr(lib, "orc_kcf_" + f).argtypes = [C.c_void_p]
    lib.orc_kcf_new.restype = C.c_void_p
    lib.orc_kcf_new.argtypes = [C.c_void_p, C.c_int]
    lib.orc_kcf_delete.argtypes = [C.c_void_p]
    lib.orc_kcf_predict.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_kcf_update.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_kalman_new.restype = C.c_void_p
    lib.orc_kalman_new.argtypes = [C.c_void_p]
    lib.orc_kalman_predict.argtypes = [C.c_void_p, C.c_void_p]
    lib.orc_kalman_update.argtypes = [C.c_void_p, C.c_void_p]
    lib.orc_kalman_get_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_kalman_delete.argtypes = [C.c_void_p]
    lib.orc_acos_table.restype = FP
    lib.orc_sse_rcp.restype = C.c_float
    lib.orc_sse_rcp.argtypes = [C.c_float]
    lib.orc_sse_rsqrt.restype = C.c_float
    lib.orc_sse_rsqrt.argtypes = [C.c_float]
    lib.orc_mot_new.restype = C.c_void_p
    lib.orc_mot_new.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.orc_mot_delete.argtypes = [C.c_void_p]
    lib.orc_mot_step.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 5
    lib.orc_mot_ntracks.argtypes = [C.c_void_p]
    lib.orc_mot_kcf.restype = C.c_void_p
    lib.orc_mot_kcf.argtypes = [C.c_void_p, C.c_int]
    lib.orc_mot_kalman.restype = C.c_void_p
    lib.orc_mot_kalman.argtypes = [C.c_void_p, C.c_int]
    _orc = lib
    return lib


def ref_available():
    return all(os.path.exists(os.path.join(REF_DIR, f)) for f in
               ["libref_hog.so", "libref_hungarian.so", "libref_drawlib.so", "libref_kalman.so", "libref_kcf.so"])


def load_ref(name):
    os.environ.setdefault("MKL_NUM_THREADS", "1")
    lib = C.CDLL(os.path.join(REF_DIR, f"libref_{name}.so"))
    if name == "kcf":
        for f in ["response", "alpha", "xm", "xf", "yf", "features", "labels", "coswin"]:
            getattr(lib, "refkcf_" + f).restype = FP
            getattr(lib, "refkcf_" + f).argtypes = [C.c_void_p]
        lib.refkcf_new.restype = C.c_void_p
        lib.refkcf_new.argtypes = [C.c_void_p]
        lib.refkcf_predict.argtypes = [C.c_void_p] * 3
        lib.refkcf_update.argtypes = [C.c_void_p] * 3
        lib.refkcf_delete.argtypes = [C.c_void_p]
        lib.refkcf_frows.argtypes = [C.c_void_p]
        lib.refkcf_fcols.argtypes = [C.c_void_p]
    if name == "kalman":
        lib.refkal_new.restype = C.c_void_p
        lib.refkal_new.argtypes = [C.c_void_p]
        lib.refkal_predict.argtypes = [C.c_void_p] * 2
        lib.refkal_update.argtypes = [C.c_void_p] * 2
        lib.refkal_state.argtypes = [C.c_void_p] * 3
        lib.refkal_delete.argtypes = [C.c_void_p]
    if name == "hog":
        lib.refhog_acos_table.restype = FP
    return lib


def arr(p, n):
    return np.ctypeslib.as_array(p, shape=(n,)).copy()


def boxes_array(boxes):
    out = np.zeros(len(boxes), BBOX_DTYPE)
    for i, b in enumerate(boxes):
        b = tuple(b)
        out[i] = (b[0], b[1], b[2], b[3], b[4] if len(b) > 4 else 0, b[5] if len(b) > 5 else 0.9)
    return out


# ---- convenience wrappers over the oracle ----
def fhog(lib, patch, h, w, mode=0):
    I = np.ascontiguousarray(patch, np.float32).ravel()
    H = np.zeros(32 * (h // 4) * (w // 4), np.float32)
    lib.orc_fhog(P(I), h, w, P(H), mode)
    return H


def crop_patch(lib, frame, box, rows, cols):
    b = boxes_array([box])
    hs, ws = abs(box[2] - box[1]) + 1, abs(box[3] - box[0]) + 1
    scratch = np.zeros(max(hs * ws, 1), np.float32)
    dst = np.zeros(rows * cols, np.float32)
    lib.orc_crop_patch(P(dst), P(scratch), P(frame), P(b), rows, cols)
    return dst


def cost_matrix(lib, trk, det):
    t, d = boxes_array(trk) if not isinstance(trk, np.ndarray) else trk, boxes_array(det) if not isinstance(det, np.ndarray) else det
    out = np.zeros(len(t) * len(d), np.float64)
    lib.orc_cost_matrix(P(t), len(t), P(d), len(d), P(out))
    return out


def assignment_optimal(lib, dist, nr, nc):
    d = np.ascontiguousarray(dist, np.float64)
    a = np.full(max(nr, 1), -1, np.int32)
    c = C.c_double(0)
    lib.orc_assignment_optimal(P(a), C.byref(c), P(d), nr, nc)
    return a[:nr], c.value


class OracleMot:
    """orc_mot_* frame loop (top/td.cpp:306-748 restated)"""

    def __init__(self, lib, kind, mode=0, cap=256):
        self.lib, self.cap, self.kind = lib, cap, kind
        self.h = C.c_void_p(lib.orc_mot_new(kind, mode, cap))

    def step(self, frame, dets):
        d = boxes_array(dets) if not isinstance(dets, np.ndarray) else dets
        cap = self.cap + 1
        pred = np.zeros(cap, BBOX_DTYPE); at = np.zeros(cap, np.int32); live = np.zeros(cap, BBOX_DTYPE); tids = np.zeros(cap, np.uint32)
        nb = C.c_int(0)
        nl = self.lib.orc_mot_step(self.h, P(frame) if frame is not None else None, P(d), len(d), P(pred), P(at), C.byref(nb), P(live), P(tids))
        return dict(predicted=pred[:nb.value].copy(), assigned=at[:nb.value].copy(), live=live[:nl].copy(), tids=tids[:nl].copy())

    def kcf(self, i):
        return C.c_void_p(self.lib.orc_mot_kcf(self.h, i))

    def kalman(self, i):
        return C.c_void_p(self.lib.orc_mot_kalman(self.h, i))

    def close(self):
        if self.h:
            self.lib.orc_mot_delete(self.h)
            self.h = None


def ref_frame_loop(kind, scene, n_frames, libs, timing=None):
    """The tracker thread body of top/td.cpp:344-644, transcribed, driving the REFERENCE's
    own per-object functions (tracker_*, assignmentoptimal, rgb2Gray, bilinearInterpolationGray)."""
    kcf, kal, hung, draw = libs
    tracks = []   # dict(h, bbox(list l,t,b,r,type,score), age, vis, inv, rows, cols, tid)
    next_tid = 0
    trace = []
    gray = np.zeros(1280 * 720, np.float32); scratch = np.zeros(1280 * 720, np.float32)

    def crop(frame, bb, rows, cols):
        l, t, b, r = bb[0], bb[1], bb[2], bb[3]
        draw.rgb2Gray(P(scratch), P(frame), l, t, r, b)
        draw.bilinearInterpolationGray(P(gray), P(scratch), b - t + 1, r - l + 1, rows, cols)

    def mk(bb):
        return BBox(int(bb[0]), int(bb[1]), int(bb[2]), int(bb[3]), int(bb[4]), float(bb[5]))

    import time as _time
    for frame, dets in scene.frames(n_frames):
        _t0 = _time.perf_counter()
        nT, nD = len(tracks), len(dets)
        pred = []
        for t in tracks:
            pb = mk(t["bbox"])
            if kind == 0:
                crop(frame, t["bbox"], t["rows"], t["cols"])
                kcf.refkcf_predict(t["h"], P(gray), C.byref(pb))
            else:
                kal.refkal_predict(t["h"], C.byref(pb))
            bb = [min(max(pb.l, 0), 1279), min(max(pb.t, 0), 719), min(max(pb.b, 0), 719), min(max(pb.r, 0), 1279), pb.type, pb.score]
            t["bbox"] = bb
            pred.append(tuple(bb[:5]))
        at = [-1] * nT; ad = [-1] * nD
        if nT and nD:
            tb = boxes_array([tuple(t["bbox"]) for t in tracks]); db = boxes_array(dets)
            # td.cpp:386-457 through the ORACLE's restatement: td.cpp itself cannot be built here (its OpenCV / yolo3.dll / pthreads-win32
            # libraries are absent and stand-ins are ruled out), so these lines are pinned by reading only -- DESIGN.md section 6
            dist = cost_matrix(load_oracle(), tb, db)
            if nT < nD:
                a = np.zeros(nT, np.int32); c = C.c_double(0)
                hung.refhung_assign(P(a), C.byref(c), P(dist), nT, nD)
                for i in range(nT):
                    at[i] = int(a[i]); ad[int(a[i])] = i
            else:
                a = np.zeros(nD, np.int32); c = C.c_double(0)
                hung.refhung_assign(P(a), C.byref(c), P(dist), nD, nT)
                for j in range(nD):
                    at[int(a[j])] = j; ad[j] = int(a[j])
        for i, t in enumerate(tracks):
            j = at[i]
            if j < 0:
                continue
            db = mk(dets[j])
            if kind == 0:
                crop(frame, dets[j], t["rows"], t["cols"])
                kcf.refkcf_update(t["h"], P(gray), C.byref(db))
            else:
                kal.refkal_update(t["h"], C.byref(db))
            t["bbox"] = list(dets[j]); t["vis"] += 1; t["age"] += 1; t["inv"] = 0
        for i, t in enumerate(tracks):
            if at[i] >= 0:
                continue
            t["age"] += 1; t["inv"] += 1
            ob = mk(t["bbox"])
            if kind == 0:
                crop(frame, t["bbox"], t["rows"], t["cols"])
                kcf.refkcf_update(t["h"], P(gray), C.byref(ob))
            else:
                kal.refkal_update(t["h"], C.byref(ob))
        keep = []
        for t in tracks:
            lost = (t["age"] < 10 and t["vis"] * 5 < 3 * t["age"]) or t["inv"] >= 20
            if not lost:
                keep.append(t)
            else:
                (kcf.refkcf_delete if kind == 0 else kal.refkal_delete)(t["h"])
        tracks = keep
        for j in range(nD):
            if ad[j] >= 0:
                continue
            d = dets[j]
            t = dict(bbox=list(d), age=0, vis=0, inv=0, rows=d[2] - d[1] + 1, cols=d[3] - d[0] + 1, tid=next_tid)
            next_tid += 1
            db = mk(d)
            if kind == 0:
                t["h"] = C.c_void_p(kcf.refkcf_new(C.byref(db)))
                draw.rgb2Gray(P(gray), P(frame), d[0], d[1], d[3], d[2])
                kcf.refkcf_update(t["h"], P(gray), C.byref(db))
            else:
                t["h"] = C.c_void_p(kal.refkal_new(C.byref(db)))
            tracks.append(t)
        if timing is not None:
            timing.append((_time.perf_counter() - _t0, nT))
        trace.append(dict(pred=pred, assigned=list(at), live=[tuple(t["bbox"][:5]) for t in tracks], tids=[t["tid"] for t in tracks]))
    for t in tracks:
        (kcf.refkcf_delete if kind == 0 else kal.refkal_delete)(t["h"])
    return trace


