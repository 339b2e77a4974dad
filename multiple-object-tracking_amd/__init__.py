"""multiple-object-tracking_amd -- host-side mirror of the MI355X tracker ABI.

Thin ctypes binding of ``libmot_amd.so`` (include/mot_abi.h).  The names follow
the reference's tracker interface (``tracker_new / tracker_predict /
tracker_update / tracker_delete / assignmentoptimal``, top/td.cpp:229-234) and
its tracker-thread loop (td.cpp:306-748).  There is no CPU implementation in
this package: if the HIP library is missing or no GPU is present every call
raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MOT_AMD_LIB") or os.path.join(_HERE, "libmot_amd.so")   # MOT_AMD_LIB: the probe build of tools/kcf_ablate.py, never the product
DROPIN_KCF_PATH = os.path.join(_HERE, "libmot_dropin_kcf.so")
DROPIN_KALMAN_PATH = os.path.join(_HERE, "libmot_dropin_kalman.so")

FRAME_W, FRAME_H = 1280, 720
TRACKER_KCF, TRACKER_KALMAN = 0, 1
FHOG_INTEL_APPROX, FHOG_EXACT = 0, 1
FFT_AUTO, FFT_GENERIC = 0, 1


class BBox(C.Structure):
    """bbox_t, top/cnntype.h:36-41 (field order l,t,b,r)."""
    _fields_ = [("l", C.c_int), ("t", C.c_int), ("b", C.c_int), ("r", C.c_int), ("type", C.c_int), ("score", C.c_float)]

    def tup(self):
        return (self.l, self.t, self.b, self.r, self.type)


BBOX_DTYPE = np.dtype([("l", "<i4"), ("t", "<i4"), ("b", "<i4"), ("r", "<i4"), ("type", "<i4"), ("score", "<f4")])


class MotConfig(C.Structure):
    _fields_ = [("device", C.c_int), ("tracker_kind", C.c_int), ("fhog_mode", C.c_int), ("fft_mode", C.c_int),
                ("max_tracks", C.c_int), ("max_dets", C.c_int), ("rank", C.c_int), ("world", C.c_int),
                ("stream", C.c_void_p), ("dev_rows", C.c_int), ("dev_cols", C.c_int), ("dev_size_lo", C.c_int), ("dev_size_hi", C.c_int), ("reserved", C.c_int * 2)]


class MotError(RuntimeError):
    pass


_lib = None


def load_library():
    """Loads libmot_amd.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MotError(f"{LIB_PATH} not built -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    lib.mot_last_error.restype = C.c_char_p
    lib.mot_ctx_stream.restype = C.c_void_p
    lib.mot_ctx_stream.argtypes = [C.c_void_p]
    _lib = lib
    return lib


def _vp(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if isinstance(a, int):
        return C.c_void_p(a)
    return a


def boxes_array(boxes) -> np.ndarray:
    """list of (l,t,b,r[,type[,score]]) / structured array -> contiguous BBOX_DTYPE array"""
    if isinstance(boxes, np.ndarray) and boxes.dtype == BBOX_DTYPE:
        return np.ascontiguousarray(boxes)
    out = np.zeros(len(boxes), BBOX_DTYPE)
    for i, b in enumerate(boxes):
        if isinstance(b, BBox):
            out[i] = (b.l, b.t, b.b, b.r, b.type, b.score)
        else:
            b = tuple(b)
            out[i] = (b[0], b[1], b[2], b[3], b[4] if len(b) > 4 else 0, b[5] if len(b) > 5 else 0.9)
    return out


class MotContext:
    """One tracker context = one GPU (one process per GPU in multi-GPU runs)."""

    def __init__(self, tracker_kind=TRACKER_KCF, device=0, max_tracks=256, max_dets=128, fhog_mode=FHOG_INTEL_APPROX,
                 fft_mode=FFT_AUTO, rank=0, world=1, stream=None, dev_size=80, dev_sizes=None):
        self.lib = load_library()
        cfg = MotConfig()
        self.lib.mot_config_default(C.byref(cfg))
        cfg.device, cfg.tracker_kind, cfg.max_tracks, cfg.max_dets = device, tracker_kind, max_tracks, max_dets
        cfg.fhog_mode, cfg.fft_mode, cfg.rank, cfg.world = fhog_mode, fft_mode, rank, world
        cfg.stream = stream
        cfg.dev_rows = cfg.dev_cols = dev_size
        if dev_sizes:                                                  # (lo, hi): one pool per square template size, per-track sizes in the device loop
            cfg.dev_size_lo, cfg.dev_size_hi = int(dev_sizes[0]), int(dev_sizes[1])
        self.cfg = cfg
        self._h = C.c_void_p()
        self._chk(self.lib.mot_ctx_create(C.byref(cfg), C.byref(self._h)))
        self.max_tracks, self.max_dets, self.world, self.rank = max_tracks, max_dets, world, rank

    def _chk(self, rc):
        if rc != 0:
            raise MotError(f"mot error {rc}: {self.lib.mot_last_error().decode()}")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.mot_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self) -> int:
        return int(self.lib.mot_ctx_stream(self._h) or 0)

    def sync(self):
        self._chk(self.lib.mot_ctx_sync(self._h))

    # ---- frame ----
    def frame_upload(self, frame_bgr: np.ndarray):
        f = np.ascontiguousarray(frame_bgr, dtype=np.uint8)
        assert f.shape == (FRAME_H, FRAME_W, 3)
        self._chk(self.lib.mot_frame_upload(self._h, _vp(f)))

    def frame_bind_device(self, dev_ptr: int):
        self._chk(self.lib.mot_frame_bind_device(self._h, C.c_void_p(dev_ptr)))

    # ---- batch stages ----
    def tracks_new(self, boxes, first_update=True) -> np.ndarray:
        b = boxes_array(boxes)
        ids = np.zeros(len(b), np.int32)
        fn = self.lib.mot_tracks_new if first_update else self.lib.mot_tracks_new_nofirst
        self._chk(fn(self._h, _vp(b), len(b), _vp(ids)))
        return ids

    def predict_batch(self, ids, clamp=True, boxes_inout=None) -> np.ndarray:
        ids = np.ascontiguousarray(ids, np.int32)
        out = boxes_array(boxes_inout).copy() if boxes_inout is not None else np.zeros(len(ids), BBOX_DTYPE)
        self._chk(self.lib.mot_predict_batch(self._h, _vp(ids), len(ids), _vp(out), int(clamp)))
        return out

    def update_batch(self, ids, boxes):
        ids = np.ascontiguousarray(ids, np.int32)
        b = boxes_array(boxes)
        self._chk(self.lib.mot_update_batch(self._h, _vp(ids), len(ids), _vp(b)))

    def delete_batch(self, ids):
        ids = np.ascontiguousarray(ids, np.int32)
        self._chk(self.lib.mot_delete_batch(self._h, _vp(ids), len(ids)))

    def _patch_ptrs(self, patches):
        keep = [np.ascontiguousarray(p, np.float32) for p in patches]
        arr = (C.c_void_p * len(keep))(*[p.ctypes.data for p in keep])
        return keep, arr

    def predict_batch_patches(self, ids, patches, boxes_inout=None) -> np.ndarray:
        ids = np.ascontiguousarray(ids, np.int32)
        keep, arr = self._patch_ptrs(patches)
        out = boxes_array(boxes_inout).copy() if boxes_inout is not None else np.zeros(len(ids), BBOX_DTYPE)
        self._chk(self.lib.mot_predict_batch_patches(self._h, _vp(ids), len(ids), arr, _vp(out)))
        return out

    def update_batch_patches(self, ids, patches, boxes):
        ids = np.ascontiguousarray(ids, np.int32)
        keep, arr = self._patch_ptrs(patches)
        b = boxes_array(boxes)
        self._chk(self.lib.mot_update_batch_patches(self._h, _vp(ids), len(ids), arr, _vp(b)))

    # ---- association ----
    def assign(self, trk, det, want_cost=True):
        t, d = boxes_array(trk), boxes_array(det)
        at = np.full(max(len(t), 1), -1, np.int32)
        ad = np.full(max(len(d), 1), -1, np.int32)
        cost = C.c_double(0.0)
        self._chk(self.lib.mot_assign(self._h, _vp(t), len(t), _vp(d), len(d), _vp(at), _vp(ad), C.byref(cost) if want_cost else None))
        return at[:len(t)], ad[:len(d)], cost.value

    def assignment_optimal(self, dist: np.ndarray, n_rows: int, n_cols: int):
        """assignmentoptimal(): dist is column-major float64 [n_rows x n_cols]"""
        d = np.ascontiguousarray(dist, np.float64).ravel()
        assert d.size == n_rows * n_cols
        a = np.full(max(n_rows, 1), -1, np.int32)
        cost = C.c_double(0.0)
        self._chk(self.lib.mot_assignment_optimal(self._h, _vp(a), C.byref(cost), _vp(d), n_rows, n_cols))
        return a[:n_rows], cost.value

    def cost_matrix(self, trk, det) -> np.ndarray:
        t, d = boxes_array(trk), boxes_array(det)
        out = np.zeros(len(t) * len(d), np.float64)
        self._chk(self.lib.mot_cost_matrix(self._h, _vp(t), len(t), _vp(d), len(d), _vp(out)))
        return out

    # ---- frame loop ----
    def step_frame(self, dets):
        d = boxes_array(dets)
        cap = self.max_tracks + 1
        pred = np.zeros(cap, BBOX_DTYPE); at = np.zeros(cap, np.int32); live = np.zeros(cap, BBOX_DTYPE); tids = np.zeros(cap, np.uint32)
        nb, nl = C.c_int(0), C.c_int(0)
        self._chk(self.lib.mot_step_frame(self._h, _vp(d), len(d), _vp(pred), _vp(at), C.byref(nb), _vp(live), _vp(tids), C.byref(nl)))
        return dict(predicted=pred[:nb.value].copy(), assigned=at[:nb.value].copy(), live=live[:nl.value].copy(), tids=tids[:nl.value].copy())

    def step_begin(self):
        ptr, spr = C.c_void_p(), C.c_int(0)
        self._chk(self.lib.mot_step_begin(self._h, C.byref(ptr), C.byref(spr)))
        return int(ptr.value or 0), spr.value

    def step_finish(self, gathered_dev_ptr, dets):
        d = boxes_array(dets)
        cap = self.max_tracks + 1
        pred = np.zeros(cap, BBOX_DTYPE); at = np.zeros(cap, np.int32); live = np.zeros(cap, BBOX_DTYPE); tids = np.zeros(cap, np.uint32)
        nb, nl = C.c_int(0), C.c_int(0)
        self._chk(self.lib.mot_step_finish(self._h, C.c_void_p(gathered_dev_ptr) if gathered_dev_ptr else None, _vp(d), len(d),
                                           _vp(pred), _vp(at), C.byref(nb), _vp(live), _vp(tids), C.byref(nl)))
        return dict(predicted=pred[:nb.value].copy(), assigned=at[:nb.value].copy(), live=live[:nl.value].copy(), tids=tids[:nl.value].copy())

    # device-resident steady-state loop
    def step_frame_device(self, frame_dev: int, dets_dev: int, n_dets: int):
        self._chk(self.lib.mot_step_frame_device(self._h, C.c_void_p(frame_dev), C.c_void_p(dets_dev), n_dets))

    def step_frame_device_ahead(self, frame_dev: int, dets_dev: int, n_dets: int, next_frame_dev: int, next_dets_dev: int, next_n_dets: int):
        """mot_step_frame_device with one frame of look-ahead (next_* = 0: none)"""
        self._chk(self.lib.mot_step_frame_device_ahead(self._h, C.c_void_p(frame_dev), C.c_void_p(dets_dev), n_dets,
                                                       C.c_void_p(next_frame_dev) if next_frame_dev else None, C.c_void_p(next_dets_dev) if next_dets_dev else None, next_n_dets))

    def step_begin_device(self, frame_dev: int):
        ptr, spr = C.c_void_p(), C.c_int(0)
        self._chk(self.lib.mot_step_begin_device(self._h, C.c_void_p(frame_dev), C.byref(ptr), C.byref(spr)))
        return int(ptr.value or 0), spr.value

    def step_begin_device_ahead(self, frame_dev: int, dets_dev: int, n_dets: int, next_frame_dev: int, next_dets_dev: int, next_n_dets: int):
        """sharded step, first half, with the frame's detection list and the NEXT frame announced (look-ahead of the detection features)"""
        ptr, spr = C.c_void_p(), C.c_int(0)
        self._chk(self.lib.mot_step_begin_device_ahead(self._h, C.c_void_p(frame_dev), C.c_void_p(dets_dev), n_dets, C.c_void_p(next_frame_dev), C.c_void_p(next_dets_dev),
                                                       next_n_dets, C.byref(ptr), C.byref(spr)))
        return int(ptr.value or 0), spr.value

    def step_finish_device(self, gathered_dev: int, dets_dev: int, n_dets: int):
        self._chk(self.lib.mot_step_finish_device(self._h, C.c_void_p(gathered_dev), C.c_void_p(dets_dev), n_dets))

    def step_frame_host(self, host_frame_ptr: int, host_dets_ptr: int, n_dets: int):
        """frame + detections from (pinned) host memory: upload on the context's copy stream, double-buffered against the previous frame"""
        self._chk(self.lib.mot_step_frame_host(self._h, C.c_void_p(host_frame_ptr), C.c_void_p(host_dets_ptr), n_dets))

    def step_frame_sharded(self, frame_dev: int, dets_dev: int, n_dets: int, nccl_comm: int):
        """one sharded frame as a single native call: predict, in-place ncclAllGather (RCCL) on the context's stream, association, update"""
        self._chk(self.lib.mot_step_frame_sharded(self._h, C.c_void_p(frame_dev), C.c_void_p(dets_dev), n_dets, C.c_void_p(nccl_comm)))

    def profile_frame_device(self, frame_dev: int, dets_dev: int, n_dets: int) -> np.ndarray:
        ms = np.zeros(5, np.float32)
        self._chk(self.lib.mot_profile_frame_device(self._h, C.c_void_p(frame_dev), C.c_void_p(dets_dev), n_dets, _vp(ms)))
        return ms

    def live_count(self) -> int:
        n = C.c_int(0)
        self._chk(self.lib.mot_live_count(self._h, C.byref(n)))
        return n.value

    def live_tracks(self):
        cap = self.max_tracks + 1
        boxes = np.zeros(cap, BBOX_DTYPE); tids = np.zeros(cap, np.uint32); ages = np.zeros(cap, np.int32); n = C.c_int(0)
        self._chk(self.lib.mot_live_tracks(self._h, _vp(boxes), _vp(tids), _vp(ages), C.byref(n)))
        return boxes[:n.value].copy(), tids[:n.value].copy(), ages[:n.value].copy()

    # ---- detector post-processing (detectors/yolo3.cpp:141-356, 490-547) ----
    def yolo_postprocess(self, head_ptrs, tensor_h, tensor_w, num_classes, image_h, image_w, obj_thresh, nms_thresh, anchors, dets_dev, cap, n_dev, want_chain=False):
        """head_ptrs: device pointers of the three raw output tensors; dets_dev / n_dev: device pointers for the boxes and their count.
        Returns the reference's bbox_chain_t (nbox, boxes) when want_chain (synchronises), else None."""
        class Opt(C.Structure):
            _fields_ = [("obj_thresh", C.c_float), ("nms_thresh", C.c_float), ("anchors", C.c_int * 18)]

        class Chain(C.Structure):
            _fields_ = [("nbox", C.c_int), ("bbox", BBox * 128)]
        o = Opt(); o.obj_thresh, o.nms_thresh = obj_thresh, nms_thresh
        for i, a in enumerate(anchors):
            o.anchors[i] = int(a)
        ch = Chain() if want_chain else None
        self._chk(self.lib.mot_yolo_postprocess(self._h, C.c_void_p(head_ptrs[0]), C.c_void_p(head_ptrs[1]), C.c_void_p(head_ptrs[2]), tensor_h, tensor_w,
                                                num_classes, image_h, image_w, C.byref(o), C.c_void_p(dets_dev), cap, C.c_void_p(n_dev),
                                                C.byref(ch) if want_chain else None))
        if not want_chain:
            return None
        out = np.zeros(ch.nbox, BBOX_DTYPE)
        for i in range(ch.nbox):
            b = ch.bbox[i]; out[i] = (b.l, b.t, b.b, b.r, b.type, b.score)
        return out

    def yolo_status(self) -> int:
        """candidates dropped for lack of workspace since the context was created (synchronises); raises MotError (MOT_ERR_CAPACITY) if any"""
        n = C.c_int(0)
        self._chk(self.lib.mot_yolo_status(self._h, C.byref(n)))
        return n.value

    # ---- overlay (td.cpp:647-733) ----
    def overlay_draw(self, frame_dev: int, boxes, tids):
        b = boxes_array(boxes); t = np.ascontiguousarray(tids, np.uint32)
        self._chk(self.lib.mot_overlay_draw(self._h, C.c_void_p(frame_dev), _vp(b), _vp(t), len(b)))

    def overlay_live(self, frame_dev: int):
        self._chk(self.lib.mot_overlay_live(self._h, C.c_void_p(frame_dev)))

    def live_response(self, live_index: int) -> np.ndarray:
        """response map of the i-th live track of the device-resident loop (most recent predict)"""
        fr, fc = C.c_int(0), C.c_int(0)
        self._chk(self.lib.mot_live_response(self._h, int(live_index), None, C.byref(fr), C.byref(fc)))
        out = np.zeros(fr.value * fc.value, np.float32)
        self._chk(self.lib.mot_live_response(self._h, int(live_index), _vp(out), C.byref(fr), C.byref(fc)))
        return out

    def live_model(self, live_index: int):
        """(xm complex64 [31 * bins], alpha float32 [bins], pos, scale (h, v), first_update, pending_det) of the i-th live track of the device loop"""
        fr, fc = C.c_int(0), C.c_int(0)
        self._chk(self.lib.mot_live_response(self._h, int(live_index), None, C.byref(fr), C.byref(fc)))
        bins = fc.value * (fr.value // 2 + 1)
        xm = np.zeros(31 * bins * 2, np.float32); al = np.zeros(bins, np.float32); pos = np.zeros(1, BBOX_DTYPE); sc = np.zeros(2, np.float32)
        first, pend = C.c_int(0), C.c_int(0)
        self._chk(self.lib.mot_live_model(self._h, int(live_index), _vp(xm), _vp(al), _vp(pos), _vp(sc), C.byref(first), C.byref(pend)))
        return xm.view(np.complex64), al, pos[0], sc, first.value, pend.value

    # ---- introspection ----
    def get_response(self, tid: int) -> np.ndarray:
        fr, fc = C.c_int(0), C.c_int(0)
        self._chk(self.lib.mot_get_response(self._h, int(tid), None, C.byref(fr), C.byref(fc)))
        out = np.zeros(fr.value * fc.value, np.float32)
        self._chk(self.lib.mot_get_response(self._h, int(tid), _vp(out), C.byref(fr), C.byref(fc)))
        return out

    def get_model(self, tid: int):
        fr, fc = C.c_int(0), C.c_int(0)
        self._chk(self.lib.mot_get_response(self._h, int(tid), None, C.byref(fr), C.byref(fc)))
        nbins = fc.value * (fr.value // 2 + 1)
        xm = np.zeros(31 * nbins * 2, np.float32); alpha = np.zeros(nbins, np.float32)
        self._chk(self.lib.mot_get_model(self._h, int(tid), _vp(xm), _vp(alpha)))
        return xm, alpha

    def get_kalman_state(self, tid: int):
        x = np.zeros(6); P = np.zeros(36)
        self._chk(self.lib.mot_get_kalman_state(self._h, int(tid), _vp(x), _vp(P)))
        return x, P

    def get_pos(self, tid: int):
        b = np.zeros(1, BBOX_DTYPE)
        self._chk(self.lib.mot_get_pos(self._h, int(tid), _vp(b)))
        return b[0]

    def debug_kcf_phases(self, enable=True):
        a = np.zeros(8, np.int64); b = np.zeros(8, np.int64)
        self._chk(self.lib.mot_debug_kcf_phases(self._h, int(enable), _vp(a), _vp(b)))
        return a, b

    def assoc_stats(self) -> np.ndarray:
        out = np.zeros(16, np.int32)
        self._chk(self.lib.mot_get_assoc_stats(self._h, _vp(out)))
        return out

    def state_save(self) -> np.ndarray:
        """checkpoint of the device-resident loop as one uint8 record (mot_state_save)"""
        n = C.c_size_t(0)
        self._chk(self.lib.mot_state_save(self._h, None, C.c_size_t(0), C.byref(n)))
        buf = np.zeros(n.value, np.uint8)
        self._chk(self.lib.mot_state_save(self._h, _vp(buf), C.c_size_t(n.value), None))
        return buf

    def state_load(self, record: np.ndarray):
        """resume from a record of state_save() (fresh context of the same configuration)"""
        rec = np.ascontiguousarray(record, np.uint8)
        self._chk(self.lib.mot_state_load(self._h, _vp(rec), C.c_size_t(rec.size)))

    def debug_trace(self):
        """(MOT_TRACE=1) int32 [2][16][max_tracks][8]: predict / update records of the last 16 frames by (frame & 15, slot); None when tracing is off"""
        n = C.c_size_t(0)
        self._chk(self.lib.mot_debug_trace_read(self._h, None, C.c_size_t(0), C.byref(n)))
        if n.value == 0:
            return None
        out = np.zeros(n.value, np.int32)
        self._chk(self.lib.mot_debug_trace_read(self._h, _vp(out), C.c_size_t(n.value), None))
        return out.reshape(2, 16, -1, 8)

    def debug_snapshot_bytes(self) -> int:
        n = C.c_size_t(0)
        self._chk(self.lib.mot_debug_snapshot(self._h, None, C.byref(n)))
        return int(n.value)

    def debug_snapshot(self, dst_dev: int):
        """stream-ordered copy of the device loop's small state into device memory at dst_dev (debug_snapshot_bytes() bytes); no synchronisation"""
        self._chk(self.lib.mot_debug_snapshot(self._h, C.c_void_p(dst_dev), None))

    def debug_profile_stages(self, enable: bool, read: bool = False):
        """two-call / sharded step: per-rank stage times of the last profiled frame [predict, all-gather, chain, residual update] in ms"""
        out = np.zeros(4, np.float32)
        self._chk(self.lib.mot_debug_profile_stages(self._h, 1 if enable else 0, _vp(out) if read else None))
        return out if read else None

    def debug_predict_timing(self, n_pairs: int):
        self._chk(self.lib.mot_debug_predict_timing(self._h, int(n_pairs)))

    def debug_predict_times(self, cap: int = 256) -> np.ndarray:
        """durations (ms) of the predict launches since debug_predict_timing() armed them, measured while the ordinary step calls ran"""
        out = np.zeros(cap, np.float32); n = C.c_int(0)
        self._chk(self.lib.mot_debug_predict_times(self._h, _vp(out), cap, C.byref(n)))
        return out[:n.value].copy()

    def assoc_trace(self, n: int = 8192) -> np.ndarray:
        """(debug, MOT_MK_TIMING=1) thread 0's (tag, 10 ns ticks) along the sparse emulation's cycles of the most recent launch"""
        out = np.zeros(n, np.int64)
        self._chk(self.lib.mot_debug_assoc_trace(self._h, _vp(out), n))
        k = int(out[0])
        v = out[1:max(1, min(k, n))]
        return np.stack([v >> 56, v & ((1 << 56) - 1)], axis=1)

    def lap_stats(self) -> np.ndarray:
        """assignment fast path: [0..7] most recent launch (outcome, rounds, free rows, searches, commits, near-tight
        edges, cyclic nodes, solver ticks), [15] what decided it (0 certificate, 1 sparse emulation, 2 dense emulation), [16..20] cumulative
        launches by outcome (0 = certified, 4 = tied optima), [26..30] dense solver: settled columns, free rows, ticks, launches it ran in,
        launches then certified"""
        out = np.zeros(32, np.int32)
        self._chk(self.lib.mot_get_lap_stats(self._h, _vp(out)))
        return out

    def fhog_extract(self, patch: np.ndarray, h: int, w: int, windowed=False) -> np.ndarray:
        p = np.ascontiguousarray(patch, np.float32).ravel()
        assert p.size == h * w
        out = np.zeros(32 * (h // 4) * (w // 4), np.float32)
        self._chk(self.lib.mot_fhog_extract(self._h, _vp(p), h, w, _vp(out), int(windowed)))
        return out

    def crop_patch(self, box, rows: int, cols: int) -> np.ndarray:
        b = boxes_array([box])
        out = np.zeros(rows * cols, np.float32)
        self._chk(self.lib.mot_crop_patch(self._h, _vp(b), rows, cols, _vp(out)))
        return out

    # ---- timers ----
    def timer_create(self, n):
        self._chk(self.lib.mot_timer_create(self._h, n))

    def timer_record(self, i):
        self._chk(self.lib.mot_timer_record(self._h, i))

    def timer_elapsed_ms(self, a, b) -> float:
        ms = C.c_float(0)
        self._chk(self.lib.mot_timer_elapsed_ms(self._h, a, b, C.byref(ms)))
        return ms.value
