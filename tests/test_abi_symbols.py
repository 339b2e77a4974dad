"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports
every symbol include/*.h declares; the per-object libraries export the
reference's mangled names; without a GPU the library fails loudly (no fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mot_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mot_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(mot):
    lib = mot.load_library()
    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"libmot_amd.so does not export {n}"


REFERENCE_MANGLED = [
    "_Z11tracker_newP11_bbox_pos_s",            # void* tracker_new(bbox_t*)             td.cpp:232
    "_Z15tracker_predictPvPfP11_bbox_pos_s",    # void tracker_predict(void*,float*,bbox_t*)  td.cpp:229
    "_Z14tracker_updatePvPfP11_bbox_pos_s",     # void tracker_update(void*,float*,bbox_t*)   td.cpp:230
    "_Z14tracker_deletePv",                     # void tracker_delete(void*)              td.cpp:231
    "_Z17assignmentoptimalPiPdS0_ii",           # void assignmentoptimal(int*,double*,double*,int,int)  td.cpp:234
]


@pytest.mark.parametrize("which", ["DROPIN_KCF_PATH", "DROPIN_KALMAN_PATH"])
def test_dropin_exports_reference_symbols(mot, which):
    path = getattr(mot, which)
    assert os.path.exists(path), f"{path} not built"
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    for sym in REFERENCE_MANGLED:
        assert re.search(rf"\bT {re.escape(sym)}\b", out), f"{os.path.basename(path)} lacks {sym}"


def test_bbox_layout_matches_reference(mot):
    # top/cnntype.h:36-41: {int l,t,b,r; int type; float score} = 24 bytes
    assert C.sizeof(mot.BBox) == 24
    assert [f[0] for f in mot.BBox._fields_] == ["l", "t", "b", "r", "type", "score"]
    assert mot.BBOX_DTYPE.itemsize == 24


def test_no_cpu_fallback_without_gpu(mot):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(mot.MotError, match="no HIP device"):
        mot.MotContext()


def test_product_does_not_reference_oracle():
    """the shipped library and host mirror must not link, import or open anything under oracle/"""
    pkg = os.path.join(ROOT, "multiple-object-tracking_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp", ".inc")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "mot_oracle" not in txt and "oracle/" not in txt.replace("tools/", ""), f"{f} references the oracle"
    out = subprocess.check_output(["ldd", os.path.join(pkg, "libmot_amd.so")], text=True)
    assert "oracle" not in out


def test_dense_solver_arming_state_machine(mot):
    """host-side scheduling of the dense LAP solver's launches (assoc_kernels.hip: dense_arming_step), no GPU needed: a constant
    detection count never arms it; a count change arms it for 8 launches; a clean stream whose count keeps changing backs off
    exponentially; a launch that needed it (device hint bit 1) holds it for 512 launches whatever the back-off says"""
    import numpy as np
    lib = mot.load_library()
    lib.mot_debug_dense_arming.argtypes = [C.c_void_p, C.c_int]
    lib.mot_debug_dense_arming.restype = C.c_int
    step = lambda h, nd: lib.mot_debug_dense_arming(h.ctypes.data_as(C.c_void_p), nd)
    h = np.zeros(16, np.int32)
    assert all(step(h, 1024) == 0 for _ in range(100))                       # the bench stream
    assert step(h, 1000) == 1 and sum(step(h, 1000) for _ in range(20)) == 6  # 8 launches (one already taken, one spent by the arming call itself)
    h[:] = 0
    armed = [step(h, 900 + (f % 7)) for f in range(4000)]                    # a clean stream whose count changes every frame
    assert sum(armed[:40]) >= 30 and sum(armed[2000:]) < 0.2 * 2000, (sum(armed[:40]), sum(armed[2000:]))
    assert h[5] >= 3                                                          # backed off several times
    h[0] = 2                                                                  # the device reports: the last launch needed the dense solver
    assert step(h, 905) == 1 and h[1] == 512 and h[4] == 0 and h[5] == 0
    h[0] = 0
    assert sum(step(h, 905) for _ in range(600)) == 511                       # held for 512 launches, then released


def test_no_null_stream_fills_in_the_product():
    """hipMemset() of device memory is asynchronous to the host and runs on the NULL stream; the contexts' streams are non-blocking and do not
    synchronise with it (tools/memset_order_probe.hip: the fill lands behind a later kernel of a non-blocking stream in 20 of 20 runs).  Round 5
    traced the rounds-4/5 parity flake to exactly that: every fill in the product is a hipMemsetAsync on the context's own stream, and
    this test keeps it so (the debug poisoning of DevBuf, followed by a device-wide wait, is the one exception)."""
    import re
    csrc = os.path.join(ROOT, "multiple-object-tracking_amd", "csrc")
    offenders = []
    for fn in sorted(os.listdir(csrc)):
        if not fn.endswith((".hip", ".h", ".cpp")):
            continue
        for ln, line in enumerate(open(os.path.join(csrc, fn)), 1):
            code = line.split("//")[0]
            if re.search(r"\bhipMemset\s*\(", code) and "poison_byte()" not in code:
                offenders.append(f"{fn}:{ln}")
            # round-5 verdict: the same hazard under other names -- the synchronous-looking fill variants (they run on the null stream too), and any
            # *Async fill / copy / launch that names the null stream explicitly (stream argument 0, nullptr, hipStreamDefault / hipStreamLegacy)
            if re.search(r"\bhipMemset(D8|D16|D32|2D|3D)\s*\(", code):
                offenders.append(f"{fn}:{ln} (null-stream fill variant)")
            if re.search(r"\bhipMem(set|cpy)\w*Async\s*\(.*,\s*(0|nullptr|NULL|hipStreamDefault|hipStreamLegacy)\s*\)\s*\)?\s*;", code):
                offenders.append(f"{fn}:{ln} (async operation on the null stream)")
            if re.search(r"hipLaunchKernelGGL\s*\([^;]*,\s*(0|nullptr|NULL)\s*,\s*[a-zA-Z_]", code) and re.search(r"dim3\([^)]*\)\s*,\s*dim3\([^)]*\)\s*,\s*[^,]+,\s*(0|nullptr|NULL)\s*,", code):
                offenders.append(f"{fn}:{ln} (kernel launch on the null stream)")
    assert not offenders, offenders
