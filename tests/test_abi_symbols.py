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
