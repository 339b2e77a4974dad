"""GPU tests of the device-resident frame loop (mot_step_frame_device): lifecycle, spawn, compaction and
sharding executed on device must reproduce the oracle's tracker-thread loop (top/td.cpp:306-748)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import orc

pytestmark = pytest.mark.gpu


def bnp(b):
    return np.stack([b[k] for k in ("l", "t", "b", "r", "type")], axis=1).reshape(-1, 5)


def _dev(frames, dets, mot):
    fd = torch.from_numpy(np.stack(frames)).cuda()
    nmax = max(len(d) for d in dets)
    da = np.zeros((len(dets), max(nmax, 1)), mot.BBOX_DTYPE)
    for i, d in enumerate(dets):
        da[i, :len(d)] = mot.boxes_array(d)
    dd = torch.from_numpy(da.view(np.uint8).reshape(len(dets), -1)).cuda()
    return fd, dd, da


@pytest.mark.parametrize("kind,n,size,nframes", [(0, 48, 80, 9), (1, 16, 80, 40), (0, 20, 64, 6), (0, 20, 96, 5)])
def test_device_loop_vs_oracle(mot, oracle, kind, n, size, nframes):
    from multiple_object_tracking_amd import synth
    scene = synth.Scene(n, size, stream_id=21 + kind, miss_pct=8, fp_pct=4)
    items = list(scene.frames(nframes))
    frames = [f for f, _ in items]; dets = [d for _, d in items]
    fd, dd, da = _dev(frames, dets, mot)
    c = mot.MotContext(tracker_kind=kind, max_tracks=128, max_dets=128, dev_size=size)
    m = orc.OracleMot(oracle, kind, 0, 128)
    for f in range(nframes):
        c.step_frame_device(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]))
        ref = m.step(frames[f], dets[f])
        boxes, tids, ages = c.live_tracks()
        diag = lambda: f"frame {f}: lap {c.lap_stats()[:16].tolist()} assoc {c.assoc_stats()[:4].tolist()} differing {[(i, bnp(boxes)[i].tolist(), bnp(ref['live'])[i].tolist()) for i in range(min(len(boxes), len(ref['live']))) if not np.array_equal(bnp(boxes)[i], bnp(ref['live'])[i])][:4]}"
        assert np.array_equal(tids, ref["tids"]), "tids: " + diag()
        assert np.array_equal(bnp(boxes), bnp(ref["live"])), "live boxes: " + diag()
    m.close(); c.close()


@pytest.mark.parametrize("n,nframes,miss,fp,size", [(600, 7, 4, 3, 80), (1024, 7, 4, 3, 80), (1024, 9, 0, 0, 80), (400, 5, 3, 2, 64)])
def test_device_loop_large_vs_oracle(mot, oracle, n, nframes, miss, fp, size):
    """the BENCHED configuration against the oracle: assignment fast path (certificate), sparse order-exact emulation, dense
    emulation as the last resort, lifecycle tail and split update.  With misses and false positives tracks die and spawn every
    frame (td.cpp:585-644) and a false positive's partner is a far-away free track: the candidate lists cannot decide those
    frames and the dense emulation runs; without them (the bench stream itself) the first two always suffice."""
    from multiple_object_tracking_amd import synth
    scene = synth.Scene(n, size, stream_id=5 if miss else 0, miss_pct=miss, fp_pct=fp)
    items = list(scene.frames(nframes))
    frames = [f for f, _ in items]; dets = [d[:1024] for _, d in items]
    fd, dd, da = _dev(frames, dets, mot)
    c = mot.MotContext(max_tracks=1024, max_dets=1024, dev_size=size)   # (64 px: the predict launch's paired workgroups and its blend split at a second template size)
    m = orc.OracleMot(oracle, 0, 0, 1024)
    used = set()
    for f in range(nframes):
        c.step_frame_device(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]))
        ref = m.step(frames[f], dets[f])
        boxes, tids, ages = c.live_tracks()
        assert np.array_equal(tids, ref["tids"]), f"frame {f} tids"
        assert np.array_equal(bnp(boxes), bnp(ref["live"])), f"frame {f} live boxes"
        if f:
            used.add(int(c.lap_stats()[15]))                          # 0 certificate, 1 sparse emulation, 2 dense emulation
    if not miss:
        assert used == {0, 1}, used                                   # the bench stream: certified frames and tie frames, never the dense emulation
    m.close(); c.close()


def test_device_loop_multiscale_150(mot, oracle):
    """BASELINE configs[4] shape on one GPU: 256 tracks spawned with a 148 x 148 template (150 px boxes would give 37.5 cells; the
    reference uses floor(size / 4) cells), then detections of 120..180 px every frame -- each update crops the detection box and
    resizes it to the track's template (td.cpp:528-537, drawlib.c:542-637), HBM-slab KCF kernels, device loop"""
    from multiple_object_tracking_amd import synth
    n, size, nframes = 256, 148, 5
    scene = synth.Scene(n, size, stream_id=44, det_sizes=(120, 180), first_frame_exact=True)
    items = list(scene.frames(nframes))
    frames = [f for f, _ in items]; dets = [d for _, d in items]
    fd, dd, da = _dev(frames, dets, mot)
    c = mot.MotContext(max_tracks=256, max_dets=256, dev_size=size)
    m = orc.OracleMot(oracle, 0, 0, 256)
    for f in range(nframes):
        c.step_frame_device(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]))
        ref = m.step(frames[f], dets[f])
        boxes, tids, ages = c.live_tracks()
        diag = lambda: f"frame {f}: lap {c.lap_stats()[:16].tolist()} assoc {c.assoc_stats()[:4].tolist()} differing {[(i, bnp(boxes)[i].tolist(), bnp(ref['live'])[i].tolist()) for i in range(min(len(boxes), len(ref['live']))) if not np.array_equal(bnp(boxes)[i], bnp(ref['live'])[i])][:4]}"
        assert np.array_equal(tids, ref["tids"]), "tids: " + diag()
        assert np.array_equal(bnp(boxes), bnp(ref["live"])), "live boxes: " + diag()
    m.close(); c.close()


@pytest.mark.parametrize("n,lo,hi,miss,fp", [(200, 72, 88, 3, 2), (96, 140, 156, 0, 0)])
def test_device_loop_per_track_template_sizes(mot, oracle, n, lo, hi, miss, fp):
    """per-track template sizes in the device-resident loop: every tracker freezes rows / cols at spawn from the spawning detection
    (kcf.cpp:148-152, td.cpp:626-627), and every frame's detections have a random size in lo..hi, so tracks of 17 different
    template sizes live side by side and each update resizes a lo..hi crop to its own track's template (td.cpp:528-537).
    One pool per size, predict / update kernels pick the descriptor per item.  72..88: LDS-resident kernels; 140..156: HBM slab."""
    from multiple_object_tracking_amd import synth
    nframes = 6
    scene = synth.Scene(n, (lo + hi) // 2, stream_id=45, det_sizes=(lo, hi), miss_pct=miss, fp_pct=fp)
    items = list(scene.frames(nframes))
    frames = [f for f, _ in items]; dets = [d[:256] for _, d in items]
    fd, dd, da = _dev(frames, dets, mot)
    c = mot.MotContext(max_tracks=256, max_dets=256, dev_sizes=(lo, hi))
    m = orc.OracleMot(oracle, 0, 0, 256)
    for f in range(nframes):
        c.step_frame_device(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]))
        ref = m.step(frames[f], dets[f])
        boxes, tids, ages = c.live_tracks()
        assert np.array_equal(tids, ref["tids"]), f"frame {f} tids"
        assert np.array_equal(bnp(boxes), bnp(ref["live"])), f"frame {f} live boxes"
    assert len(tids) >= n // 2
    m.close(); c.close()


def test_device_loop_eight_ranks_on_one_gpu(mot, oracle):
    """BASELINE configs[3] shape: 1024 tracks sharded over eight contexts (here on one GPU, the all-gather emulated with device copies) must
    equal the unsharded oracle.  Round 5: a spawning track goes to the least loaded rank, so a segment holds ceil(1024 / 8) = 128 boxes
    (3 KB) -- with tracks dying and spawning every frame (2 % misses, 1 % false positives) no rank may overflow it (a sticky device
    error of live_tracks() otherwise)"""
    from multiple_object_tracking_amd import synth
    hip = C.CDLL("libamdhip64.so")
    W, n = 8, 1024
    scene = synth.Scene(n, 80, stream_id=6, miss_pct=2, fp_pct=1)
    items = list(scene.frames(6))
    frames = [f for f, _ in items]; dets = [d[:1024] for _, d in items]
    fd, dd, da = _dev(frames, dets, mot)
    ranks = [mot.MotContext(max_tracks=1024, max_dets=1024, rank=r, world=W) for r in range(W)]
    m = orc.OracleMot(oracle, 0, 0, 1024)
    for f in range(len(frames)):
        segs = [c.step_begin_device(fd[f].data_ptr()) for c in ranks]
        spr = segs[0][1]
        assert spr == 128
        for c in ranks:
            c.sync()
        bases = [segs[r][0] - r * spr * 24 for r in range(W)]
        for dst in range(W):
            for src in range(W):
                if src != dst:
                    assert hip.hipMemcpy(C.c_void_p(bases[dst] + src * spr * 24), C.c_void_p(segs[src][0]), spr * 24, 3) == 0
        for r, c in enumerate(ranks):
            c.step_finish_device(bases[r], dd[f].data_ptr(), len(dets[f]))
        ref = m.step(frames[f], dets[f])
        for r, c in enumerate(ranks):
            boxes, tids, _ = c.live_tracks()
            assert np.array_equal(tids, ref["tids"]), f"frame {f} rank {r}"
            assert np.array_equal(bnp(boxes), bnp(ref["live"])), f"frame {f} rank {r}"
    m.close()
    for c in ranks:
        c.close()


@pytest.mark.parametrize("n,ahead", [(30, False), (30, True), (700, True)])
def test_device_loop_sharded_two_ranks(mot, oracle, n, ahead):
    """two ranks on one GPU, the all-gather emulated by copies: both reproduce the unsharded oracle.  ahead: the two-call form with
    the detection list and the NEXT frame announced (mot_step_begin_device_ahead) -- at 700 tracks the next frame's detection features are
    then computed one frame early on the side stream of every rank"""
    from multiple_object_tracking_amd import synth
    hip = C.CDLL("libamdhip64.so")
    scene = synth.Scene(n, 80, stream_id=31, miss_pct=6, fp_pct=4)
    items = list(scene.frames(7))
    frames = [f for f, _ in items]; dets = [d for _, d in items]
    fd, dd, da = _dev(frames, dets, mot)
    cap = 64 if n <= 60 else 1024
    ranks = [mot.MotContext(max_tracks=cap, max_dets=cap, rank=r, world=2) for r in range(2)]
    m = orc.OracleMot(oracle, 0, 0, cap)
    for f in range(len(frames)):
        if ahead:
            nxt = f + 1 if f + 1 < len(frames) else None
            segs = [c.step_begin_device_ahead(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]), fd[nxt].data_ptr() if nxt is not None else 0,
                                              dd[nxt].data_ptr() if nxt is not None else 0, len(dets[nxt]) if nxt is not None else 0) for c in ranks]
        else:
            segs = [c.step_begin_device(fd[f].data_ptr()) for c in ranks]
        spr = segs[0][1]
        for c in ranks:
            c.sync()
        bases = [segs[r][0] - r * spr * 24 for r in range(2)]
        for dst in range(2):                      # emulate ncclAllGather: every rank receives every segment
            for src in range(2):
                if src != dst:
                    assert hip.hipMemcpy(C.c_void_p(bases[dst] + src * spr * 24), C.c_void_p(segs[src][0]), spr * 24, 3) == 0
        for r, c in enumerate(ranks):
            c.step_finish_device(bases[r], dd[f].data_ptr(), len(dets[f]))
        ref = m.step(frames[f], dets[f])
        for r, c in enumerate(ranks):
            boxes, tids, _ = c.live_tracks()
            assert np.array_equal(tids, ref["tids"]), f"frame {f} rank {r}"
            assert np.array_equal(bnp(boxes), bnp(ref["live"])), f"frame {f} rank {r}"
    m.close()
    for c in ranks:
        c.close()


def test_device_loop_steady_1024(mot):
    """BASELINE configs[2] shape: 1024 tracks stay alive and keep tracking their objects for a few frames."""
    from multiple_object_tracking_amd import synth
    scene = synth.Scene(1024, 80, stream_id=0)
    items = list(scene.frames(6))
    frames = [f for f, _ in items]; dets = [d for _, d in items]
    fd, dd, da = _dev(frames, dets, mot)
    c = mot.MotContext(max_tracks=1024, max_dets=1024)
    for f in range(len(frames)):
        c.step_frame_device(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]))
    boxes, tids, ages = c.live_tracks()
    assert len(boxes) == 1024 and set(tids.tolist()) == set(range(1024))
    assert np.all(ages == 5)
    # every track sits on a detection of the last frame
    last = {tuple(int(v) for v in d[:4]) for d in dets[-1]}
    assert all(tuple(int(v) for v in bnp(boxes)[i][:4]) in last for i in range(1024))
    c.close()


def test_device_loop_fused_update_subprocess():
    """MOT_SPLIT_UPDATE=0 keeps the single fused update kernel (features + model update in one launch, no side stream);
    the default splits it into detection features beside the association and a blend.  Both must reproduce the oracle:
    the switch is read when the device loop is created, so the fused variant runs in a child process."""
    import subprocess, sys
    env = dict(os.environ, MOT_SPLIT_UPDATE="0")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(orc.ROOT, "tests", "test_gpu_devloop.py"), "-q", "-x", "-m", "gpu",
                          "-k", "test_device_loop_vs_oracle or test_device_loop_sharded_two_ranks"],
                         cwd=orc.ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("kind", [0, 1])
def test_device_loop_empty_and_bursty_frames(mot, oracle, kind):
    """frames without any detection (every track coasts on its predicted box until the lost rule removes it, td.cpp:589),
    a burst of new detections, then an empty stream again: the device loop (split update, fused lifecycle) must follow the
    oracle's tracker thread through all of it"""
    from multiple_object_tracking_amd import synth
    n, size, nframes = 24, 80, 34
    scene = synth.Scene(n, size, stream_id=77 + kind, miss_pct=5, fp_pct=5)
    items = list(scene.frames(nframes))
    frames = [f for f, _ in items]; dets = [list(d) for _, d in items]
    for f in (3, 4, 11):                       # isolated empty frames
        dets[f] = []
    for f in range(14, nframes):               # the detector goes silent: all tracks are lost after 20 invisible frames
        dets[f] = []
    dets[12] = dets[12] + dets[12][: n // 2]   # duplicated boxes: more detections than tracks, ties in the cost matrix
    fd, dd, da = _dev(frames, dets, mot)
    c = mot.MotContext(tracker_kind=kind, max_tracks=128, max_dets=128, dev_size=size)
    m = orc.OracleMot(oracle, kind, 0, 128)
    for f in range(nframes):
        c.step_frame_device(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]))
        ref = m.step(frames[f], dets[f])
        boxes, tids, ages = c.live_tracks()
        assert np.array_equal(tids, ref["tids"]), f"frame {f} tids"
        assert np.array_equal(bnp(boxes), bnp(ref["live"])), f"frame {f} live boxes"
    assert len(ref["tids"]) == 0, "every track must have been dropped by the lost rule"
    m.close(); c.close()


def test_device_loop_per_track_sizes_two_ranks_on_one_gpu(mot, oracle):
    """per-track template sizes AND the shard over two ranks together: two contexts of one GPU (the all-gather emulated with device
    copies), 150 tracks of sizes 74..86 with misses and false positives, must equal the unsharded oracle"""
    from multiple_object_tracking_amd import synth
    hip = C.CDLL("libamdhip64.so")
    W, n, lo, hi = 2, 150, 74, 86
    scene = synth.Scene(n, 80, stream_id=46, det_sizes=(lo, hi), miss_pct=3, fp_pct=2)
    items = list(scene.frames(6))
    frames = [f for f, _ in items]; dets = [d[:256] for _, d in items]
    fd, dd, da = _dev(frames, dets, mot)
    ranks = [mot.MotContext(max_tracks=256, max_dets=256, rank=r, world=W, dev_sizes=(lo, hi)) for r in range(W)]
    m = orc.OracleMot(oracle, 0, 0, 256)
    for f in range(len(frames)):
        segs = [c.step_begin_device(fd[f].data_ptr()) for c in ranks]
        spr = segs[0][1]
        for c in ranks:
            c.sync()
        bases = [segs[r][0] - r * spr * 24 for r in range(W)]
        for dst in range(W):
            for src in range(W):
                if src != dst:
                    assert hip.hipMemcpy(C.c_void_p(bases[dst] + src * spr * 24), C.c_void_p(segs[src][0]), spr * 24, 3) == 0
        for r, c in enumerate(ranks):
            c.step_finish_device(bases[r], dd[f].data_ptr(), len(dets[f]))
        ref = m.step(frames[f], dets[f])
        for r, c in enumerate(ranks):
            boxes, tids, _ = c.live_tracks()
            assert np.array_equal(tids, ref["tids"]), f"frame {f} rank {r}"
            assert np.array_equal(bnp(boxes), bnp(ref["live"])), f"frame {f} rank {r}"
    m.close()
    for c in ranks:
        c.close()


def test_device_loop_detector_noise_dense_solver(mot, oracle):
    """a stream with detector noise -- 4 % misses, 3 % false positives, no two detections on one centroid (NMS) -- forces far matches
    in almost every frame: the sparse solver's prices fail there, the dense solver (lap_dense.hip) certifies most frames, the rest go
    to the emulation.  Every frame must equal the oracle, and the dense solver must have certified some."""
    from multiple_object_tracking_amd import synth
    n, nframes = 300, 10
    scene = synth.Scene(n, 80, stream_id=47, miss_pct=4, fp_pct=3, nms=True)
    items = list(scene.frames(nframes))
    frames = [f for f, _ in items]; dets = [d[:384] for _, d in items]
    fd, dd, da = _dev(frames, dets, mot)
    c = mot.MotContext(max_tracks=384, max_dets=384)
    m = orc.OracleMot(oracle, 0, 0, 384)
    for f in range(nframes):
        c.step_frame_device(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]))
        ref = m.step(frames[f], dets[f])
        boxes, tids, ages = c.live_tracks()
        assert np.array_equal(tids, ref["tids"]), f"frame {f} tids"
        assert np.array_equal(bnp(boxes), bnp(ref["live"])), f"frame {f} live boxes"
    st = c.lap_stats()
    assert st[29] >= 3 and st[30] >= 2, st.tolist()                  # dense solver ran / was certified (the first noisy frame arms it)
    m.close(); c.close()


def test_device_loop_single_size_class_lo_equals_hi(mot, oracle):
    """dev_size_lo == dev_size_hi != dev_rows (round-2 advisor finding): one size class means the single-template path with THAT
    template -- detections of size lo must spawn and be tracked in a lo x lo template, not be dropped against dev_rows = 80"""
    from multiple_object_tracking_amd import synth
    n, size, nframes = 40, 64, 5
    scene = synth.Scene(n, size, stream_id=46, miss_pct=5, fp_pct=3)
    items = list(scene.frames(nframes))
    frames = [f for f, _ in items]; dets = [d for _, d in items]
    fd, dd, da = _dev(frames, dets, mot)
    c = mot.MotContext(max_tracks=128, max_dets=128, dev_size=80, dev_sizes=(size, size))
    m = orc.OracleMot(oracle, 0, 0, 128)
    for f in range(nframes):
        c.step_frame_device(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]))
        ref = m.step(frames[f], dets[f])
        boxes, tids, ages = c.live_tracks()
        assert len(tids) > 0 and np.array_equal(tids, ref["tids"]), f"frame {f} tids"
        assert np.array_equal(bnp(boxes), bnp(ref["live"])), f"frame {f} live boxes"
    m.close(); c.close()


def test_device_loop_response_peaks_1024(mot, oracle):
    """north_star's KCF tolerance at the BENCHED size: after 4 frames of the 1024-track stream every live track's response map
    (kcf_t::response of the last predict; deferred blend, split update and joined launches all behind it) has its peak within 1e-4
    relative of the oracle's and the same arg-max"""
    from multiple_object_tracking_amd import synth
    n, nframes = 1024, 4
    scene = synth.Scene(n, 80, stream_id=0)
    items = list(scene.frames(nframes))
    frames = [f for f, _ in items]; dets = [d for _, d in items]
    fd, dd, da = _dev(frames, dets, mot)
    c = mot.MotContext(max_tracks=1024, max_dets=1024)
    m = orc.OracleMot(oracle, 0, 0, 1024)
    for f in range(nframes):
        c.step_frame_device(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]))
        ref = m.step(frames[f], dets[f])
    boxes, tids, ages = c.live_tracks()
    assert np.array_equal(tids, ref["tids"]) and np.array_equal(bnp(boxes), bnp(ref["live"]))
    worst = 0.0
    for i in range(0, len(tids), 3):                                    # every third track: 342 response maps
        got = c.live_response(i)
        exp = orc.arr(oracle.orc_kcf_response(m.kcf(i)), got.size)
        assert int(np.argmax(got)) == int(np.argmax(exp)), f"track {i}: arg-max"
        rel = abs(float(got.max()) - float(exp.max())) / abs(float(exp.max()))
        worst = max(worst, rel)
    assert worst <= 1e-4, worst
    m.close(); c.close()


@pytest.mark.parametrize("n,cap,nframes", [(150, 256, 8), (600, 1024, 9)])
def test_device_loop_fed_from_host_memory(mot, oracle, n, cap, nframes):
    """mot_step_frame_host: frames and detection lists come from pinned host memory, uploaded on the context's copy stream into three
    rotating device buffers -- several frames enqueued back to back without synchronising (so uploads run two frames ahead, buffers are
    reused, and at the larger size every frame's detection features are launched one frame early on the side stream) must still
    reproduce the oracle"""
    from multiple_object_tracking_amd import synth
    scene = synth.Scene(n, 80, stream_id=33, miss_pct=6, fp_pct=4)
    items = list(scene.frames(nframes))
    frames = [f for f, _ in items]; dets = [d for _, d in items]
    nmax = max(len(d) for d in dets)
    da = np.zeros((nframes, max(nmax, 1)), mot.BBOX_DTYPE)
    for i, d in enumerate(dets):
        da[i, :len(d)] = mot.boxes_array(d)
    pf = torch.from_numpy(np.stack(frames)).pin_memory()
    pd = torch.from_numpy(da.view(np.uint8).reshape(nframes, -1)).pin_memory()
    c = mot.MotContext(max_tracks=cap, max_dets=cap)
    m = orc.OracleMot(oracle, 0, 0, cap)
    refs = [m.step(frames[f], dets[f]) for f in range(nframes)]
    for f in range(nframes):                                            # all frames enqueued; read back only at the end and in the middle
        c.step_frame_host(pf[f].data_ptr(), pd[f].data_ptr(), len(dets[f]))
        if f in (3, nframes - 1):
            boxes, tids, ages = c.live_tracks()
            assert np.array_equal(tids, refs[f]["tids"]), f"frame {f} tids"
            assert np.array_equal(bnp(boxes), bnp(refs[f]["live"])), f"frame {f} live boxes"
    m.close(); c.close()


@pytest.mark.parametrize("n,miss,fp", [(1024, 0, 0), (300, 6, 4), (48, 8, 5)])
def test_device_loop_lookahead_vs_oracle(mot, oracle, n, miss, fp):
    """mot_step_frame_device_ahead: the next frame's detection features are computed one frame early (three spectra buffers in
    rotation, deferred blend reading the previous frame's).  Results must equal the oracle frame by frame -- also when the look-ahead is
    wrong (a different frame / list arrives than was announced), missing (end of stream), or alternates with plain calls."""
    from multiple_object_tracking_amd import synth
    nframes = 9
    scene = synth.Scene(n, 80, stream_id=7, miss_pct=miss, fp_pct=fp)
    items = list(scene.frames(nframes))
    frames = [f for f, _ in items]; dets = [d[:1024] for _, d in items]
    fd, dd, da = _dev(frames, dets, mot)
    c = mot.MotContext(max_tracks=1024, max_dets=1024)
    m = orc.OracleMot(oracle, 0, 0, 1024)
    for f in range(nframes):
        nxt = f + 1 if f + 1 < nframes else None
        if f == 3:
            c.step_frame_device(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]))              # plain call in between (ignores what was prefetched for it? no: uses it)
        elif f == 5 and nxt is not None:
            c.step_frame_device_ahead(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]), fd[0].data_ptr(), dd[0].data_ptr(), len(dets[0]))   # announces the WRONG next frame
        else:
            c.step_frame_device_ahead(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]),
                                      fd[nxt].data_ptr() if nxt is not None else 0, dd[nxt].data_ptr() if nxt is not None else 0, len(dets[nxt]) if nxt is not None else 0)
        ref = m.step(frames[f], dets[f])
        boxes, tids, ages = c.live_tracks()
        diag = lambda: f"frame {f}: lap {c.lap_stats()[:16].tolist()} assoc {c.assoc_stats()[:4].tolist()} differing {[(i, bnp(boxes)[i].tolist(), bnp(ref['live'])[i].tolist()) for i in range(min(len(boxes), len(ref['live']))) if not np.array_equal(bnp(boxes)[i], bnp(ref['live'])[i])][:4]}"
        assert np.array_equal(tids, ref["tids"]), "tids: " + diag()
        assert np.array_equal(bnp(boxes), bnp(ref["live"])), "live boxes: " + diag()
    m.close(); c.close()


def test_device_loop_one_buffer_reused_in_stream_order(mot, oracle):
    """ONE device frame buffer and ONE detection buffer, overwritten for every frame by copies enqueued on the context's own stream right
    behind the previous step call, nothing synchronised in between (mot_abi.h: "work the caller enqueues on mot_ctx_stream() after the
    call may overwrite both").  At 1024 tracks the detection features of a frame run on a second stream beside its association chain:
    the step call must order the context's stream behind that launch, or the next upload overwrites a frame that is still being read
    (round-3 advisor finding).  Checked against the oracle at the end and half way."""
    from multiple_object_tracking_amd import synth
    n, nframes = 1024, 8
    scene = synth.Scene(n, 80, stream_id=11)
    items = list(scene.frames(nframes))
    frames = [f for f, _ in items]; dets = [d[:1024] for _, d in items]
    src_f = torch.from_numpy(np.stack(frames)).cuda()                  # staging copies (device to device on the context's stream)
    _, src_d, _ = _dev(frames, dets, mot)
    c = mot.MotContext(max_tracks=1024, max_dets=1024)
    ext = torch.cuda.ExternalStream(c.stream)
    buf_f = torch.empty_like(src_f[0]); buf_d = torch.empty_like(src_d[0])
    m = orc.OracleMot(oracle, 0, 0, 1024)
    torch.cuda.synchronize()
    for f in range(nframes):
        with torch.cuda.stream(ext):
            buf_f.copy_(src_f[f], non_blocking=True); buf_d.copy_(src_d[f], non_blocking=True)
        c.step_frame_device(buf_f.data_ptr(), buf_d.data_ptr(), len(dets[f]))
        ref = m.step(frames[f], dets[f])
        if f in (nframes // 2, nframes - 1):
            boxes, tids, _ = c.live_tracks()
            assert np.array_equal(tids, ref["tids"]), f"frame {f} tids"
            assert np.array_equal(bnp(boxes), bnp(ref["live"])), f"frame {f} live boxes"
    m.close(); c.close()


def test_size_class_after_single_pool_use(mot, oracle):
    """a 72-px pool first created by a single-pool call (mot_fhog_extract: in-place direct transforms, no region T) and then used as a size class of the
    device loop: get_pool re-derives the layout with region T for the size-class kernels -- same results as the oracle"""
    from multiple_object_tracking_amd import synth
    lo, hi, n, nframes = 72, 76, 40, 4
    c = mot.MotContext(max_tracks=64, max_dets=64, dev_sizes=(lo, hi))
    rng = np.random.default_rng(3)
    I = rng.uniform(0, 255, size=(72, 72)).astype(np.float32)
    H = c.fhog_extract(I, 72, 72)
    assert np.array_equal(H.view(np.uint32), orc.fhog(oracle, I, 72, 72, 0).view(np.uint32))
    scene = synth.Scene(n, (lo + hi) // 2, stream_id=46, det_sizes=(lo, hi))
    items = list(scene.frames(nframes))
    frames = [f for f, _ in items]; dets = [d[:64] for _, d in items]
    fd, dd, da = _dev(frames, dets, mot)
    m = orc.OracleMot(oracle, 0, 0, 64)
    for f in range(nframes):
        c.step_frame_device(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]))
        ref = m.step(frames[f], dets[f])
        boxes, tids, ages = c.live_tracks()
        assert np.array_equal(tids, ref["tids"]), f"frame {f} tids"
        assert np.array_equal(bnp(boxes), bnp(ref["live"])), f"frame {f} live boxes"
    m.close(); c.close()


def _state_hashes(env_extra, args):
    import subprocess, sys
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "state_dump.py")] + [str(a) for a in args],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    return [ln for ln in out.stdout.splitlines() if ln.startswith("frame ")]


@pytest.mark.parametrize("args", [(48, 128, 8, 4, 9, 21), (30, 32, 30, 0, 8, 33), (300, 1024, 6, 4, 6, 7, "--ahead")])
def test_folded_geometry_kernels_bit_equal_general_kernels(args):
    """Round-4 advisor finding: the 80 x 80 px kernels with the template geometry folded in as constants (kMode 7: predict, feature, direct update
    kernels) must leave the SAME BITS in device memory as the general kernels: model, alpha, pos, scale, flags and response map of every live track, after every frame of
    noisy streams (tracks that keep their predicted box every frame; the second case has more of them than the residual-update grid has
    workgroups, so its multi-item loop runs).  One process per variant (the switch is read once)."""
    general = _state_hashes({"MOT_KCF_K80": "0"}, args)
    assert len(general) == args[4]
    for k80 in ("7", "3"):
        assert _state_hashes({"MOT_KCF_K80": k80}, args) == general, f"MOT_KCF_K80={k80} differs from the general kernels"
    # the DIRECT update kernel (every update of a frame when the blend is not deferred): its <7> instantiation holds the folded copy inlined
    if len(args) == 6:
        assert _state_hashes({"MOT_KCF_K80": "7", "MOT_DEFER_BLEND": "0"}, args) == _state_hashes({"MOT_KCF_K80": "0", "MOT_DEFER_BLEND": "0"}, args), \
            "direct update kernel: folded <7> differs from the general <1>"


@pytest.mark.parametrize("size", [72, 76])
def test_inplace_transforms_against_the_ping_pong_form(size, tmp_path):
    """Round-5 verdict (measurement hygiene): kernel mode 5 -- LDS-resident templates whose direct transforms run IN PLACE (72 / 76 px single pools: region T
    leaves the layout, two workgroups per CU) -- against the same kernels with the second buffer (MOT_DFT_INPLACE=0), over a noisy stream (tracks die, spawn
    and keep their predicted boxes).  NOT bit-equal, unlike the folded-geometry kernels: the transform code re-enables floating-point contraction (FFTW's
    own rounding is not reproducible) and the two code shapes contract differently -- measured: first difference in frame 1's responses.  The
    bars are the ones both hold against the oracle: live lists, ids, positions and flags EQUAL after every frame; models and alphas within 2e-5 of the
    largest coefficient, response peaks within 1e-4 relative with equal arg-max.  One process per variant."""
    import subprocess, sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "state_dump.py")
    out = {}
    for v in ("1", "0"):
        npz = str(tmp_path / f"inplace{v}.npz")
        r = subprocess.run([sys.executable, tool, "40", "64", "6", "4", "6", "23", "--size", str(size), "--npz", npz], env=dict(os.environ, MOT_DFT_INPLACE=v),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        out[v] = (np.load(npz), [ln.split()[:4] for ln in r.stdout.splitlines() if ln.startswith("frame ")])
    a, b = out["1"][0], out["0"][0]
    assert out["1"][1] == out["0"][1] and len(out["1"][1]) == 6                    # same live counts per frame
    assert sorted(a.files) == sorted(b.files) and len(a.files) > 100
    for k in a.files:
        x, y = a[k], b[k]
        if k.endswith("_pos") or k.endswith("_flags"):
            assert np.array_equal(x, y), k
        elif k.endswith("_resp"):
            assert int(np.argmax(x)) == int(np.argmax(y)) and abs(float(x.max()) - float(y.max())) <= 1e-4 * abs(float(y.max())) + 1e-12, k
        else:
            assert np.abs(x - y).max() <= 2e-5 * max(np.abs(y).max(), 1e-30), k


def test_sparse_view_miscompile_and_its_fix():
    """Round-4 advisor finding, closed in round 5: the sparse update body with the FOLDED descriptor (MOT_KCF_K80 bit 3, compiled only into the
    demonstration libraries of `make endcf`) writes a wrong model because hipcc places register copies in front of a folded EXEC restore
    (kcf_kernels.hip, kcf_update_sparse_run; tools/isa_exec0_scan.py).  Same source with -mllvm -amdgpu-remove-redundant-endcf=0: the bits of the
    general kernels.  The default build of it must still differ -- when a compiler release fixes it, this reminder fails and the folded
    instantiation can be reconsidered."""
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multiple-object-tracking_amd")
    view, fixed = os.path.join(pkg, "libmot_amd_view.so"), os.path.join(pkg, "libmot_amd_view_endcf.so")
    if not (os.path.exists(view) and os.path.exists(fixed)):
        pytest.skip("demonstration libraries not built (make -C multiple-object-tracking_amd/csrc endcf)")
    args = (48, 128, 8, 4, 9, 21)
    general = _state_hashes({"MOT_KCF_K80": "0"}, args)
    assert len(general) == args[4]
    assert _state_hashes({"MOT_AMD_LIB": fixed, "MOT_KCF_K80": "15"}, args) == general, "inner EXEC restores kept: the folded sparse update must equal the general kernels"
    assert _state_hashes({"MOT_AMD_LIB": fixed, "MOT_KCF_K80": "7"}, args) == general
    assert _state_hashes({"MOT_AMD_LIB": view, "MOT_KCF_K80": "15"}, args) != general, "the default build of the folded sparse update now equals the general kernels: re-evaluate MOT_KCF_K80 bit 3"


def test_finish_refuses_a_different_detection_list(mot):
    """two-call form: the detection spectra belong to the list given to mot_step_begin_device_ahead (round-4 advisor finding)"""
    from multiple_object_tracking_amd import synth
    scene = synth.Scene(24, 80, stream_id=3)
    items = list(scene.frames(3))
    fd, dd, da = _dev([f for f, _ in items], [d for _, d in items], mot)
    c = mot.MotContext(max_tracks=64, max_dets=64)
    n0 = len(items[0][1])
    c.step_begin_device_ahead(fd[0].data_ptr(), dd[0].data_ptr(), n0, 0, 0, 0)
    with pytest.raises(mot.MotError):
        c.step_finish_device(0, dd[1].data_ptr(), n0)                   # another list than the begin call's
    c.close()


def test_setup_fills_are_ordered_before_the_first_frame(mot, oracle):
    """Round 5's root cause of the "look-ahead flake" (rounds 4-5: about one run in 10^4 of the device loop left the oracle, only when nothing was
    synchronised between frames and the chip was busy): the set-up filled the pending-detection array with hipMemset(), which for device memory is
    asynchronous to the host and runs on the NULL stream -- and the context's non-blocking stream does not synchronise with the null stream
    (tools/memset_order_probe.hip shows the mechanism in isolation: 20 of 20).  When the fill executed late it wiped what the first frame's lifecycle
    step had just written, every track's FIRST model update slipped from the second frame's predict to the third's, and near-twin tracks swapped
    detections (tests/test_flake_replay.py reproduces two failing runs box for box on the CPU).  This test keeps the null stream busy in front of the
    first step call and requires the first update to have run in the second frame's predict.  (It is a sanity check, not a discriminator: in a library
    built before the fix the pool set-up's synchronous table uploads happen to drain the null stream just before the critical fill, so the late fill
    needs a busy CHIP, which only the soak provides -- profiles/r05_hunt_soak.log.  tests/test_abi_symbols.py keeps hipMemset( out of the product.)"""
    from multiple_object_tracking_amd import synth
    n = 64
    scene = synth.Scene(n, 80, stream_id=13)
    items = list(scene.frames(3))
    frames = [f for f, _ in items]; dets = [d for _, d in items]
    fd, dd, da = _dev(frames, dets, mot)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    big = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    c = mot.MotContext(max_tracks=1024, max_dets=1024)
    for _ in range(600):                                                # ~0.15 s of fills queued on the NULL stream (what tools/memset_order_probe.hip does with a spin kernel),
        assert hip.hipMemsetAsync(C.c_void_p(big.data_ptr()), 0, C.c_size_t(1 << 30), None) == 0   # right in front of the first step: the device loop's set-up runs inside it
    c.step_frame_device(fd[0].data_ptr(), dd[0].data_ptr(), len(dets[0]))      # spawns the tracks: pend_det[slot] = detection, first_update = 1
    torch.cuda.synchronize()                                            # the null stream has drained: a fill that was queued on it has landed by now -- behind frame 0
    c.step_frame_device(fd[1].data_ptr(), dd[1].data_ptr(), len(dets[1]))      # its predict must run the first update
    boxes, tids, _ = c.live_tracks()
    m = orc.OracleMot(oracle, 0, 0, 1024)
    m.step(frames[0], dets[0]); ref = m.step(frames[1], dets[1])
    assert np.array_equal(tids, ref["tids"]) and np.array_equal(bnp(boxes), bnp(ref["live"]))
    for i in range(0, len(tids), 7):
        xm, al, pos, sc, first, pend = c.live_model(i)
        assert first == 0, f"track {i}: the first model update did not run in the second frame's predict (pending detection {pend})"
        assert np.abs(xm).max() > 0 and np.abs(al).max() > 0
    del big
    m.close(); c.close()


def _full_state(c):
    boxes, tids, ages = c.live_tracks()
    parts = [bnp(boxes).tobytes(), tids.tobytes(), ages.tobytes()]
    for i in range(len(tids)):
        xm, al, pos, sc, first, pend = c.live_model(i)
        if first:                                                       # never predicted: the slot still holds its previous owner's model
            xm = np.zeros_like(xm); al = np.zeros_like(al); resp = np.zeros(1, np.float32)
        else:
            resp = c.live_response(i)
        parts += [xm.tobytes(), al.tobytes(), pos.tobytes(), sc.tobytes(), resp.tobytes(), bytes([first & 255, pend & 255, (pend >> 8) & 255])]
    return b"".join(parts)


@pytest.mark.parametrize("kind,n,cap", [(0, 120, 256), (0, 300, 1024), (1, 60, 128), (0, 800, 1024)])   # 800 tracks: frames committed provisionally around the checkpoint (round 6)
def test_state_save_load_resumes_bit_for_bit(mot, oracle, kind, n, cap):
    """mot_state_save / mot_state_load (state dump / load): a stream is run for 5 frames, checkpointed into a host record and
    resumed in a FRESH context; frames 5..9 of the resumed loop must give the oracle's live lists, and -- KCF -- leave the same bits in device memory
    as the uninterrupted run (models, alphas, positions, flags, response maps of every live track).  Noisy streams: tracks die, spawn and keep their
    predicted boxes across the checkpoint; the pending (deferred) model updates and the spectra they refer to travel with the record."""
    from multiple_object_tracking_amd import synth
    scene = synth.Scene(n, 80, stream_id=31 + kind, miss_pct=5, fp_pct=3)
    items = list(scene.frames(10))
    frames = [f for f, _ in items]; dets = [d[:cap] for _, d in items]
    fd, dd, da = _dev(frames, dets, mot)
    m = orc.OracleMot(oracle, kind, 0, cap)
    refs = [m.step(frames[f], dets[f]) for f in range(10)]
    m.close()

    def run(c, f0, f1):
        for f in range(f0, f1):
            c.step_frame_device(fd[f].data_ptr(), dd[f].data_ptr(), len(dets[f]))
            boxes, tids, _ = c.live_tracks()
            assert np.array_equal(tids, refs[f]["tids"]) and np.array_equal(bnp(boxes), bnp(refs[f]["live"])), f"frame {f}"
    a = mot.MotContext(tracker_kind=kind, max_tracks=cap, max_dets=cap)
    run(a, 0, 10)
    end_a = _full_state(a) if kind == 0 else None
    a.close()
    b = mot.MotContext(tracker_kind=kind, max_tracks=cap, max_dets=cap)
    run(b, 0, 5)
    record = b.state_save()
    b.close()
    c = mot.MotContext(tracker_kind=kind, max_tracks=cap, max_dets=cap)
    c.state_load(record)
    run(c, 5, 10)
    if kind == 0:
        assert _full_state(c) == end_a, "the resumed loop left different bits in device memory than the uninterrupted one"
    with pytest.raises(mot.MotError):
        c.state_load(record)                                            # not a fresh context any more
    c.close()


def test_state_save_load_sharded_two_ranks(mot, oracle):
    """... and with the tracks sharded over two ranks (two contexts on one GPU, the all-gather emulated by copies): each rank's record holds the
    replicated live list WITH the owner of every track and this rank's segment bookkeeping; both ranks are checkpointed after frame 4 under track
    churn (owners no longer round-robin), resumed in fresh contexts, and frames 5..8 must give the oracle's live lists on both.  (Round-5 advisor
    finding: the bit-for-bit test covered world == 1 only.  The association workspace -- solver statistics, the dense solver's arming hint -- is
    deliberately not part of a record: it steers which tier runs, never a result.)"""
    from multiple_object_tracking_amd import synth
    hip = C.CDLL("libamdhip64.so")
    n, cap = 90, 128
    scene = synth.Scene(n, 80, stream_id=37, miss_pct=6, fp_pct=4)
    items = list(scene.frames(9))
    frames = [f for f, _ in items]; dets = [d[:cap] for _, d in items]
    fd, dd, da = _dev(frames, dets, mot)
    m = orc.OracleMot(oracle, 0, 0, cap)
    refs = [m.step(frames[f], dets[f]) for f in range(9)]
    m.close()

    def run(ranks, f0, f1):
        for f in range(f0, f1):
            segs = [c.step_begin_device(fd[f].data_ptr()) for c in ranks]
            spr = segs[0][1]
            for c in ranks:
                c.sync()
            bases = [segs[r][0] - r * spr * 24 for r in range(2)]
            for dst in range(2):
                for src in range(2):
                    if src != dst:
                        assert hip.hipMemcpy(C.c_void_p(bases[dst] + src * spr * 24), C.c_void_p(segs[src][0]), spr * 24, 3) == 0
            for r, c in enumerate(ranks):
                c.step_finish_device(bases[r], dd[f].data_ptr(), len(dets[f]))
            for r, c in enumerate(ranks):
                boxes, tids, _ = c.live_tracks()
                assert np.array_equal(tids, refs[f]["tids"]) and np.array_equal(bnp(boxes), bnp(refs[f]["live"])), f"frame {f} rank {r}"
    a = [mot.MotContext(max_tracks=cap, max_dets=cap, rank=r, world=2) for r in range(2)]
    run(a, 0, 5)
    records = [c.state_save() for c in a]
    for c in a:
        c.close()
    b = [mot.MotContext(max_tracks=cap, max_dets=cap, rank=r, world=2) for r in range(2)]
    for c, rec in zip(b, records):
        c.state_load(rec)
    run(b, 5, 9)
    # a record belongs to its rank: the other rank's context refuses it
    fresh = mot.MotContext(max_tracks=cap, max_dets=cap, rank=0, world=2)
    with pytest.raises(mot.MotError):
        fresh.state_load(records[1])
    fresh.close()
    for c in b:
        c.close()


def test_provisional_commits_leave_the_bits_of_the_waiting_loop():
    """Round 6 (mot_dev.h: ProvRec): a tie frame whose certificate fails only because of a few disjoint two-row cycles is committed
    at once with the solver's optimum; the tracks of the cycles are cloned into shadow slots that adopt the other detection, the NEXT predict
    launch computes both alternatives beside the order-exact emulation (a kernel of its own on the emulation stream), and the patch step copies
    the shadows over the tracks of every pair the emulation reports as swapped.  The bench stream at 1024 tracks, 30 frames with one frame of
    look-ahead and NOTHING synchronised or read back before the end (the path bench.py times: patch step behind the predict, shadow items'
    predicted boxes taken over), and the same stream read back after every frame (the patch step runs at the synchronisation point, before the
    predict; 9 frames: ties in frames 1, 5 and 7): live list, ids, ages and every live track's model, alpha, position, scale, flags and response map must be the bits MOT_PROV=0
    leaves -- the loop that waits for the emulation inside the frame, as rounds 3-5 did.  (Both are held to the oracle elsewhere in this file.)"""
    args = (1024, 1024, 0, 0, 30, 0, "--ahead", "--final-only")
    import subprocess, sys
    def run(env_extra, a):
        out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "state_dump.py")] + [str(x) for x in a],
                             env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
        frames = [ln for ln in out.stdout.splitlines() if ln.startswith("frame ")]
        stats = dict(zip(*[iter([ln for ln in out.stdout.splitlines() if ln.startswith("stats ")][0].split()[1:])] * 2))
        return frames, {k: int(v) for k, v in stats.items()}
    f1, s1 = run({"MOT_PROV": "1"}, args)
    f0, s0 = run({"MOT_PROV": "0"}, args)
    assert len(f1) == 1 and f1 == f0, (f1, f0)
    assert s0["provisional"] == 0 and s1["tie"] == s0["tie"] and s1["tie"] >= 8
    assert s1["provisional"] >= 8 and s1["swaps"] >= 3 and s1["dense_bits"] == 0, s1     # this stream: every tie frame is a set of disjoint pairs, about half of them swapped
    # ... and frame by frame through the synchronisation points (patch step before the predict, no shadow boxes to take over)
    args2 = (1024, 1024, 0, 0, 9, 0, "--ahead")
    g1, t1 = run({"MOT_PROV": "1"}, args2)
    g0, _ = run({"MOT_PROV": "0"}, args2)
    assert len(g1) == 9 and g1 == g0
    assert t1["provisional"] >= 3


def test_provisional_commit_bit_decided_by_the_dense_emulation():
    """... and the rare leg: the sparse emulation of a provisionally committed frame REFUSES (an entry outside its candidate lists could have mattered),
    so the dense order-exact emulation inside the patch step decides the swap bits from the frame's own copies of the predicted boxes and the detection
    list.  Forced through the test hook MOT_PROV=2 (the patch step ignores the sparse emulation's answer); same bits as the waiting loop, unsynchronised
    and through the synchronisation points.  (tests/test_gpu_variants.py holds the same hook to the oracle under other switch combinations.)"""
    import subprocess, sys
    def run(env_extra, a):
        out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "state_dump.py")] + [str(x) for x in a],
                             env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
        frames = [ln for ln in out.stdout.splitlines() if ln.startswith("frame ")]
        stats = dict(zip(*[iter([ln for ln in out.stdout.splitlines() if ln.startswith("stats ")][0].split()[1:])] * 2))
        return frames, {k: int(v) for k, v in stats.items()}
    for a in ((1024, 1024, 0, 0, 14, 0, "--ahead", "--final-only"), (1024, 1024, 0, 0, 8, 0, "--ahead")):
        f2, s2 = run({"MOT_PROV": "2"}, a)
        f0, _ = run({"MOT_PROV": "0"}, a)
        assert f2 == f0 and len(f2) >= 1, a
        assert s2["provisional"] >= 3 and s2["dense_bits"] == s2["provisional"], s2
