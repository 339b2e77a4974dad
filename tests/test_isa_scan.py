"""The shipped code objects hold no vector instruction in front of an EXEC restore (round 5: the cause of the one wrong result the
folded sparse-update instantiation produced -- register copies the allocator placed behind a divergent loop whose own EXEC restore
had been folded into the enclosing region's; tools/isa_exec0_scan.py).  No GPU needed: llvm-objdump on the
code objects inside libmot_amd.so."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multiple-object-tracking_amd")


def _scanner():
    spec = importlib.util.spec_from_file_location("isa_exec0_scan", os.path.join(ROOT, "tools", "isa_exec0_scan.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    if not os.path.exists(mod.OBJDUMP): pytest.skip("llvm-objdump not found")
    return mod


def test_scanner_rules_on_a_synthetic_listing():
    """both rules fire on the shape of the miscompiled block and stay quiet on the regular lowering"""
    mod = _scanner()
    bad = """_Z3badv:
.LBB0_1:
	v_add_u32_e32 v1, 1, v1
	s_andn2_b64 exec, exec, s[4:5]
	s_cbranch_execnz .LBB0_1
.LBB0_2:
	v_mov_b64_e32 v[72:73], v[88:89]
	s_waitcnt lgkmcnt(0)
	s_barrier
	s_or_b64 exec, exec, s[8:9]
	s_endpgm
""".split("\n")
    hits, sites = mod.scan_lines(bad, False)
    assert sites == 1 and len(hits) == 1 and hits[0][0].startswith("loop exit") and hits[0][3][0][1].startswith("v_mov_b64")
    good = [ln for ln in bad if "v_mov_b64" not in ln]
    good.insert(good.index("\ts_endpgm"), "\tv_mov_b64_e32 v[72:73], v[88:89]")             # behind the restore: fine
    assert mod.scan_lines(good, False) == ([], 1)
    join = """_Z4joinv:
	s_and_saveexec_b64 s[4:5], vcc
	s_cbranch_execz .LBB1_2
	v_add_u32_e32 v1, 1, v1
.LBB1_2:
	v_mov_b32_e32 v2, v3
	s_or_b64 exec, exec, s[8:9]
	s_endpgm
""".split("\n")
    hits, sites = mod.scan_lines(join, False)
    assert sites == 1 and len(hits) == 1 and hits[0][0].startswith("join")
    lanes = [ln.replace("v_mov_b32_e32 v2, v3", "v_readlane_b32 s0, v2, 3") for ln in join]   # ignores EXEC: not reported
    assert mod.scan_lines(lanes, False) == ([], 1)


def test_shipped_library_is_clean():
    mod = _scanner()
    lib = os.path.join(PKG, "libmot_amd.so")
    if not os.path.exists(lib): pytest.skip("libmot_amd.so not built")
    hits, sites = mod.scan_file(lib)
    assert sites > 5000, f"only {sites} loop exits / joins found: the scan did not see the code objects"
    assert hits == [], "vector instructions in front of an EXEC restore: " + "; ".join(f"{h[1]} ({h[0]})" for h in hits)


def test_demonstration_library_is_flagged():
    """the default build of the folded sparse-update body is what the scan exists for; with the inner restores kept it is clean"""
    mod = _scanner()
    view, fixed = os.path.join(PKG, "libmot_amd_view.so"), os.path.join(PKG, "libmot_amd_view_endcf.so")
    if not (os.path.exists(view) and os.path.exists(fixed)): pytest.skip("demonstration libraries not built (make -C multiple-object-tracking_amd/csrc endcf)")
    hits, _ = mod.scan_file(view)
    assert hits and all("kcf_update_sparse_run" in h[1] and "Lb1E" in h[1] for h in hits), hits
    assert any(len(h[3]) == 7 and all(s.startswith("v_mov_b64") for _, s in h[3]) for h in hits)
    assert mod.scan_file(fixed)[0] == []


def test_hot_kernels_keep_their_register_budget():
    """What the design's occupancy figures rest on, read from the shipped code objects' metadata: the 80-px KCF
    kernels run two 512-thread workgroups per CU (<= 128 VGPRs) without scratch; the association's one-workgroup kernels have 1024 threads
    (<= 128 VGPRs) and no scratch either.  `python tools/kernel_resources.py` prints the whole table (profiles/r05_kernel_resources.txt)."""
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec); spec.loader.exec_module(kr)
    if not os.path.exists(kr.READELF): pytest.skip("llvm-readelf not found")
    lib = os.path.join(PKG, "libmot_amd.so")
    if not os.path.exists(lib): pytest.skip("libmot_amd.so not built")
    ks = kr.kernels(lib); names = kr.demangle(list(ks))
    if all(names[n] == n for n in ks): pytest.skip("c++filt not available")
    by_name = {}
    for n, d in ks.items():
        pretty = names[n].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        by_name[pretty] = d
    hot = ["kcf_predict_kernel<7>", "kcf_features_kernel<7>", "kcf_predict_features_kernel<7, false>", "kcf_update_kernel<7>", "kcf_predict_multi_kernel<7>",
           "kcf_predict_kernel<1>", "kcf_features_kernel<1>", "kcf_update_kernel<1>", "kcf_predict_multi_kernel<1>", "kcf_predict_multi_kernel<3>", "kcf_update_multi_kernel<3>",
           "lap_rowscan_kernel", "lap_solve_kernel", "lap_solve2_kernel<false>", "mk_sparse_kernel<false>", "mk_sparse_stream_kernel<false>", "munkres_kernel<false, false>", "lap_dense_kernel",
           "kalman_predict_kernel", "kalman_update_kernel"]
    for k in hot:
        assert k in by_name, f"{k} not in the library (have: {sorted(by_name)[:8]} ...)"
        d = by_name[k]
        assert d["private_segment_fixed_size"] == 0, f"{k}: {d['private_segment_fixed_size']} B/lane of scratch"
        assert d["vgpr_count"] <= 128, f"{k}: {d['vgpr_count']} VGPRs"
    assert by_name["kcf_features_kernel<7>"]["vgpr_count"] <= 88             # (round 4's figure; three workgroups per CU would need <= 84)
