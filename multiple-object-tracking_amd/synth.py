"""Deterministic synthetic camera stream for the tracker hot path.

The reference has no dataset, detector weights or recorded frames; the detector
(detectors/yolo3.cpp) stays a stub fed by synthetic boxes.  A scene is a static
smooth background plus N textured square sprites moving at constant integer
velocity and reflecting at the frame border; detections are the ground-truth
boxes with +-2 px integer jitter, class = k mod 3, shuffled.

Everything is integer arithmetic on a splitmix64 stream so the same bytes are
produced on every host (the container and the GPU box).
"""
from __future__ import annotations

import numpy as np

FRAME_W, FRAME_H = 1280, 720
MASK = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & MASK

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
        return z ^ (z >> 31)

    def randint(self, lo: int, hi: int) -> int:
        """uniform integer in [lo, hi]"""
        return lo + self.next() % (hi - lo + 1)

    def bytes(self, n: int) -> np.ndarray:
        words = np.array([self.next() for _ in range((n + 7) // 8)], dtype=np.uint64)
        return words.view(np.uint8)[:n].copy()


def _int_sine(period: int, amp: int) -> np.ndarray:
    k = np.arange(period, dtype=np.float64)
    return np.rint(amp * np.sin(2.0 * np.pi * k / period)).astype(np.int32)


class Scene:
    def __init__(self, n_objects: int, size: int = 80, stream_id: int = 0, det_sizes: tuple[int, int] | None = None,
                 miss_pct: int = 0, fp_pct: int = 0, first_frame_exact: bool = False, nms: bool = False):
        self.n, self.size = n_objects, size
        self.det_sizes = det_sizes  # (lo, hi): detection boxes get a random square size in [lo, hi] around the object centre
        # first_frame_exact: frame 0's detections have exactly `size` (every track is spawned with the size x size template,
        # td.cpp:626-627 freezes it), the random sizes start with frame 1 -> BASELINE configs[4]: one template size, multi-scale
        # detections resized to it (td.cpp:528-537)
        self.first_frame_exact = first_frame_exact
        self.miss_pct, self.fp_pct = miss_pct, fp_pct
        # nms: like a detector's non-maximum suppression, no two detections of a frame share a centroid (the later one is dropped).
        # Coinciding centroids are what makes optima TIE in this synthetic scene (integer pixel grid, 1000 objects in 1280 x 720)
        self.nms = nms
        seed = 0x5EED0000 + stream_id
        self.seed = seed
        rng = SplitMix64(seed)
        # background: three integer sinusoids, amplitude 20 each, around 128, small per-channel offsets
        s1, s2, s3 = _int_sine(257, 20), _int_sine(181, 20), _int_sine(331, 20)
        xx = np.arange(FRAME_W)[None, :]
        yy = np.arange(FRAME_H)[:, None]
        base = 128 + s1[xx % 257] + s2[yy % 181] + s3[(xx + 2 * yy) % 331]
        bg = np.stack([base - 6, base, base + 9], axis=-1)
        self.background = np.clip(bg, 0, 255).astype(np.uint8)
        # objects
        self.pos = np.zeros((n_objects, 2), np.int64)   # x, y of the top-left corner
        self.vel = np.zeros((n_objects, 2), np.int64)
        self.sprites = []
        for k in range(n_objects):
            r = SplitMix64(seed ^ ((k * 0x9E3779B97F4A7C15) & MASK))
            self.pos[k] = (r.randint(4, FRAME_W - size - 4), r.randint(4, FRAME_H - size - 4))
            vx, vy = r.randint(-3, 3), r.randint(-3, 3)
            self.vel[k] = (vx, vy)
            nblk = (size + 7) // 8
            blocks = r.bytes(nblk * nblk * 3).reshape(nblk, nblk, 3).astype(np.int32)
            tex = np.kron(blocks, np.ones((8, 8, 1), np.int32))[:size, :size]
            noise = (r.bytes(size * size * 3).reshape(size, size, 3).astype(np.int32) % 21) - 10
            self.sprites.append(np.clip(tex + noise, 0, 255).astype(np.uint8))
        self.det_rng = rng
        self.t = 0

    def gt_boxes(self) -> np.ndarray:
        """(n,4) l,t,b,r inclusive"""
        s = self.size
        return np.stack([self.pos[:, 0], self.pos[:, 1], self.pos[:, 1] + s - 1, self.pos[:, 0] + s - 1], axis=1)

    def render(self) -> np.ndarray:
        f = self.background.copy()
        s = self.size
        for k in range(self.n):
            x, y = int(self.pos[k, 0]), int(self.pos[k, 1])
            f[y:y + s, x:x + s] = self.sprites[k]
        return f

    def detections(self):
        """list of (l,t,b,r,type,score) in shuffled order; also returns the object index of each detection"""
        rng = self.det_rng
        s = self.size
        dets, owner = [], []
        for k in range(self.n):
            if self.miss_pct and rng.randint(0, 99) < self.miss_pct:
                continue
            jx, jy = rng.randint(-2, 2), rng.randint(-2, 2)
            ds = s
            if self.det_sizes and not (self.first_frame_exact and self.t == 0):
                ds = rng.randint(self.det_sizes[0], self.det_sizes[1])
            cx, cy = int(self.pos[k, 0]) + s // 2 + jx, int(self.pos[k, 1]) + s // 2 + jy
            l = min(max(cx - ds // 2, 0), FRAME_W - ds)
            t = min(max(cy - ds // 2, 0), FRAME_H - ds)
            dets.append((l, t, t + ds - 1, l + ds - 1, k % 3, 0.9))
            owner.append(k)
        if self.fp_pct:
            for _ in range(self.n):
                if rng.randint(0, 99) < self.fp_pct:
                    l, t = rng.randint(0, FRAME_W - s), rng.randint(0, FRAME_H - s)
                    dets.append((l, t, t + s - 1, l + s - 1, rng.randint(0, 2), 0.9))
                    owner.append(-1)
        if self.nms:
            seen, kd, ko = set(), [], []
            for dd, oo in zip(dets, owner):
                key = ((dd[0] + dd[3]) >> 1, (dd[1] + dd[2]) >> 1)
                if key in seen:
                    continue
                seen.add(key); kd.append(dd); ko.append(oo)
            dets, owner = kd, ko
        # Fisher-Yates shuffle
        for i in range(len(dets) - 1, 0, -1):
            j = rng.randint(0, i)
            dets[i], dets[j] = dets[j], dets[i]
            owner[i], owner[j] = owner[j], owner[i]
        return dets, owner

    def advance(self):
        s = self.size
        for k in range(self.n):
            for a, lim in ((0, FRAME_W), (1, FRAME_H)):
                p = int(self.pos[k, a]) + int(self.vel[k, a])
                if p < 2 or p > lim - s - 2:
                    self.vel[k, a] = -self.vel[k, a]
                    p = int(self.pos[k, a]) + int(self.vel[k, a])
                self.pos[k, a] = p
        self.t += 1

    def frames(self, n_frames: int):
        """yields (frame_bgr u8[720,1280,3], detections list) for n_frames consecutive frames"""
        for _ in range(n_frames):
            yield self.render(), self.detections()[0]
            self.advance()
