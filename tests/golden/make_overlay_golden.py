#!/usr/bin/env python3
"""Golden vectors of the overlay step (top/td.cpp:647-733): the REFERENCE's own drawRect (top/drawlib.c:97-151, compiled by
oracle/Makefile into oracle/_ref/libref_drawlib.so) driven through the tracker thread's drawing loop -- three nested outlines per
track in colormap[hashcolor(tid + 1) & 255] (td.cpp:619-620).  The colour table is read as DATA from the reference's td.cpp:655-697 (256 integers), the
hash (td.cpp:295-304) is restated here.  Build container only.  Output: tests/golden/overlay_cases.npz -- per case the boxes, the
track ids and the frame bytes the reference changed (flat byte offsets + values over a zero frame), plus the 256-entry table."""
import ctypes as C
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402

REF_TD = "/root/reference/top/td.cpp"


def ref_colormap():
    txt = open(REF_TD, errors="ignore").read()
    body = txt[txt.index("static uint32_t colormap[]"):]
    body = body[body.index("{") + 1:body.index("};")]
    vals = [int(v, 16) for v in re.findall(r"0x[0-9a-fA-F]{6}", body)]
    assert len(vals) == 256, len(vals)
    return np.array(vals, np.uint32)


def hashcolor(a):
    M = 0xFFFFFFFF
    a = ((a + 0x7ed55d16) + (a << 12)) & M
    a = ((a ^ 0xc761c23c) ^ (a >> 19)) & M
    a = ((a + 0x165667b1) + (a << 5)) & M
    a = ((a + 0xd3a2646c) ^ (a << 9)) & M
    a = ((a + 0xfd7046c5) + (a << 3)) & M
    a = ((a ^ 0xb55a4f09) ^ (a >> 16)) & M
    return a


def cases():
    rng = np.random.default_rng(20261003)
    out = []
    # crowded: 40 overlapping 80 x 80 boxes
    b = []
    for _ in range(40):
        l, t = int(rng.integers(0, 400)), int(rng.integers(0, 300)); b.append((l, t, t + 79, l + 79))
    out.append((b, list(range(100, 140))))
    # sizes 1..7 px, frame borders, mixed
    b = [(10, 10, 10, 10), (20, 20, 22, 22), (30, 30, 33, 33), (40, 40, 44, 44), (50, 50, 56, 56), (0, 0, 79, 79), (1200, 640, 719, 1279),
         (0, 300, 719, 5), (600, 0, 4, 1279), (5, 5, 100, 300), (5, 5, 300, 100)]
    out.append((b, [7, 8, 9, 10, 11, 3000000000, 12, 13, 14, 15, 16]))
    # 90 tracks of mixed sizes, dense
    b = []
    for _ in range(90):
        l, t = int(rng.integers(0, 1200)), int(rng.integers(0, 640)); s = int(rng.integers(60, 100)); b.append((l, t, min(t + s, 719), min(l + s, 1279)))
    out.append((b, [int(x) for x in rng.integers(0, 2 ** 32, 90)]))
    return out


def main():
    draw = orc.load_ref("drawlib")
    draw.drawRect.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint32]
    cm = ref_colormap()
    out = {"colormap": cm}
    cs = cases()
    out["n"] = np.int32(len(cs))
    for k, (boxes, tids) in enumerate(cs):
        frame = np.zeros(720 * 1280 * 3 + 4096, np.uint8)              # (slack behind the frame: the reference writes unchecked)
        sent = np.zeros_like(frame)
        for (l, t, b, r), tid in zip(boxes, tids):
            color = int(cm[hashcolor((tid + 1) & 0xFFFFFFFF) & 255])    # td.cpp:619-620: tid = tracker_id++, color = hashcolor(tracker_id): the id after the increment
            for d in range(3):                                         # td.cpp:701-731
                draw.drawRect(orc.P(frame), l + d, t + d, r - d, b - d, color)
                draw.drawRect(orc.P(sent), l + d, t + d, r - d, b - d, 0xFFFFFF)
        assert not sent[720 * 1280 * 3:].any(), "a case writes outside the frame"
        idx = np.nonzero(sent[:720 * 1280 * 3])[0].astype(np.int32)     # every byte the reference touched (black outlines included)
        out[f"boxes_{k}"] = orc.boxes_array([bb + (0, 0.9) for bb in boxes]); out[f"tids_{k}"] = np.array(tids, np.uint32)
        out[f"idx_{k}"] = idx; out[f"val_{k}"] = frame[idx]
    np.savez_compressed(os.path.join(HERE, "overlay_cases.npz"), **out)
    print("overlay_cases.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.startswith("idx")})


if __name__ == "__main__":
    main()
