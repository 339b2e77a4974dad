"""Link-level proof of "drops into top/td.cpp unchanged" (container-only: needs /root/reference).

The reference's tracker thread is compiled HERE, unmodified and by path, to an object file -- once with -DKCF_TRACKER
(td.cpp:47, the KCF build) and once without (the Kalman build, yolo3tracker.vcxproj:138).  Nothing is run and no
stand-in for OpenCV / the detector DLL is written: the test reads the undefined symbols of the object (and of a shared object linked from it) and checks that

  * the five tracker symbols it wants (td.cpp:229-234, C++ linkage, Itanium-mangled) are exactly what
    libmot_dropin_kcf.so / libmot_dropin_kalman.so export,
  * the C helpers it wants (td.cpp:236-261: rgb2Gray, bilinearInterpolationGray, drawRect; the reference compiles top/drawlib.c for
    them) are exported by the drop-in libraries as well (device kernels behind the reference's signatures), so no object of the
    reference besides td.o takes part in the link,
  * everything else it wants belongs to OpenCV, the detector DLL (tensor*), or the C/C++ runtime -- i.e. there is no
    tracker-side symbol left that the drop-in libraries fail to provide.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PKG = os.path.join(ROOT, "multiple-object-tracking_amd")

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "top", "td.cpp")), reason="reference checkout not present (GPU box)")

TRACKER_SYMS = {
    "_Z11tracker_newP11_bbox_pos_s", "_Z15tracker_predictPvPfP11_bbox_pos_s", "_Z14tracker_updatePvPfP11_bbox_pos_s",
    "_Z14tracker_deletePv", "_Z17assignmentoptimalPiPdS0_ii",
}
HELPER_SYMS = {"rgb2Gray", "bilinearInterpolationGray", "drawRect"}


def _undefined(obj):
    out = subprocess.check_output(["nm", "-u", obj], text=True)
    return {ln.split()[-1] for ln in out.splitlines() if ln.strip()}


def _exports(lib, dynamic=True):
    out = subprocess.check_output(["nm", "-D" if dynamic else "-g", "--defined-only", lib], text=True)
    return {ln.split()[-1].split("@")[0] for ln in out.splitlines() if ln.strip()}


def _runtime_exports():
    libs = []
    ldd = subprocess.check_output(["ldd", os.path.join(PKG, "libmot_dropin_kcf.so")], text=True)
    for ln in ldd.splitlines():
        m = re.search(r"=>\s*(\S+)", ln)
        if m and re.search(r"lib(c|m|stdc\+\+|gcc_s|pthread|dl|rt)\.so", m.group(1)):
            libs.append(m.group(1))
    assert libs, "could not locate the C/C++ runtime libraries"
    syms = set()
    for lib in libs:
        syms |= _exports(lib)
    return syms


@pytest.mark.parametrize("kind,defs,lib", [("kcf", ["-DKCF_TRACKER"], "libmot_dropin_kcf.so"), ("kalman", [], "libmot_dropin_kalman.so")])
def test_unmodified_td_cpp_links_against_dropin(tmp_path, kind, defs, lib):
    obj = str(tmp_path / f"td_{kind}.o")
    # the only adaptation is the reference's own (commented-out) _aligned_malloc mapping, cnntype.h:14-16; -idirafter keeps the
    # vendored pthreads-win32 headers from shadowing the system ones
    subprocess.check_call(["g++", "-O1", "-fPIC", "-std=c++11", "-w", *defs, "-include", os.path.join(ROOT, "oracle", "ref_platform.h"),
                           f"-I{REF}/top", "-idirafter", f"{REF}/include", "-c", f"{REF}/top/td.cpp", "-o", obj])
    und = _undefined(obj)
    # 1. the tracker interface: wanted by td.cpp, exported by the drop-in library, name for name
    assert TRACKER_SYMS <= und, f"td.cpp no longer references {TRACKER_SYMS - und}"
    exp = _exports(os.path.join(PKG, lib))
    assert TRACKER_SYMS <= exp, f"{lib} lacks {TRACKER_SYMS - exp}"
    # 2. the C helpers (td.cpp:235-261; the reference compiles top/drawlib.c for them) are exported by the drop-in library too, with C linkage
    want_helpers = und & HELPER_SYMS
    assert want_helpers == (HELPER_SYMS if kind == "kcf" else {"drawRect"}), want_helpers
    assert HELPER_SYMS <= exp, f"{lib} lacks {HELPER_SYMS - exp}"
    # 3. nothing tracker-side is left: the rest is OpenCV, the detector DLL, or the runtime
    rest = und - TRACKER_SYMS - HELPER_SYMS
    runtime = _runtime_exports()
    foreign = {s for s in rest if not (s.startswith(("_ZN2cv", "_ZNK2cv", "tensor")) or s in runtime
                                       or s in ("_GLOBAL_OFFSET_TABLE_", "__dso_handle"))}
    assert not foreign, f"td.cpp wants symbols nobody provides: {sorted(foreign)}"
    # 4. and the object really links WITHOUT any other object of the reference: a shared object from td.o + the drop-in library leaves only
    # OpenCV / detector symbols undefined (the running harness, harness/td_stubs.cpp + tests/test_td_harness.py, provides those)
    so = str(tmp_path / f"td_{kind}.so")
    subprocess.check_call(["g++", "-shared", "-fPIC", "-o", so, obj, f"-L{PKG}", f"-l:{lib}", f"-Wl,-rpath,{PKG}", "-lpthread"])
    left = {s.split("@")[0] for s in _undefined(so)}
    dyn = subprocess.check_output(["readelf", "-d", so], text=True)
    assert lib in dyn, "the linked object does not depend on the drop-in library"
    unresolved = {s for s in left if not (s in TRACKER_SYMS or s in HELPER_SYMS or s in runtime or s.startswith(("_ZN2cv", "_ZNK2cv", "tensor", "_ITM_", "__gmon")))}
    assert not unresolved, sorted(unresolved)


def test_harness_executables_build_from_the_unmodified_td_cpp():
    """oracle/Makefile `harness`: td.cpp (by path, both builds) + harness/td_stubs.cpp + the drop-in library link into executables with no
    undefined symbol left besides the C / C++ runtime -- in particular none of OpenCV, of the detector DLL or of drawlib.c."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "harness"], stdout=subprocess.DEVNULL)
    runtime = _runtime_exports()
    for kind, lib in (("kcf", "libmot_dropin_kcf.so"), ("kalman", "libmot_dropin_kalman.so")):
        exe = os.path.join(ROOT, "oracle", "_ref", f"td_harness_{kind}")
        assert os.path.exists(exe)
        und = {s.split("@")[0] for s in _undefined(exe)}
        exp = _exports(os.path.join(PKG, lib))
        missing = {s for s in und if not (s in exp or s in runtime or s.startswith(("_ITM_", "__gmon")))}
        assert not missing, sorted(missing)
        assert TRACKER_SYMS <= und and (und & HELPER_SYMS) == (HELPER_SYMS if kind == "kcf" else {"drawRect"})
