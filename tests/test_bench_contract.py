"""CPU checks of the measurement plumbing: algorithmic byte counts, the PMC -> traffic derivation
on the committed counter files, and the shape of the committed bench line (the contract bench.py has to keep)."""
import importlib.util
import json
import os
import subprocess
import sys

import orc

ROOT = orc.ROOT


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_bytes_match_baseline():
    bench = _load(os.path.join(ROOT, "bench.py"), "bench_mod")
    ab = bench.alg_bytes(80)
    assert ab["predict"] == 19200 + 54560 + 880 + 48                    # u8 crop + xm + alpha + pos in / box out
    assert ab["update"] == 19200 + 2 * 54560 + 2 * 880 + 24
    assert ab["predict"] + ab["update"] == 204792
    assert ab["blend"] == 3 * 54560 + 3 * 880 + 24                      # spectrum read, model read + write, alpha


def test_traffic_derivation_on_committed_counters():
    prof = os.path.join(ROOT, "profiles")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "derive_traffic.py"),
                          os.path.join(prof, "r01_pmc_fetch_size.csv"), os.path.join(prof, "r01_pmc_write_size.csv"), "1024"],
                         capture_output=True, text=True, check=True)
    tj = json.loads(out.stdout)
    committed = json.load(open(os.path.join(prof, "r01_traffic.json")))
    assert tj == committed
    alg_predict = 1024 * (19200 + 54560 + 880 + 48)
    # measured HBM traffic of the predict launch is within 5 % of the algorithmic bytes (no wasted re-reads)
    assert 1.0 <= tj["kcf_predict_bytes_per_launch_n1024"] / alg_predict < 1.05


def test_round2_traffic_derivation_and_bench_line():
    prof = os.path.join(ROOT, "profiles")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "derive_traffic.py"),
                          os.path.join(prof, "r02_pmc_fetch_size.csv"), os.path.join(prof, "r02_pmc_write_size.csv"), "1024"],
                         capture_output=True, text=True, check=True)
    tj = json.loads(out.stdout)
    assert tj == json.load(open(os.path.join(prof, "r02_traffic.json")))
    assert 1.0 <= tj["kcf_predict_bytes_per_launch_n1024"] / (1024 * (19200 + 54560 + 880 + 48)) < 1.05
    for name in ("r02_bench_n1024.json", "r02_bench_n1024_driver_style.json"):
        j = json.loads(open(os.path.join(prof, name)).read().strip().splitlines()[-1])
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                    "dtype", "data", "config", "roofline", "cpu_baseline", "steady_state", "latency_bound"):
            assert key in j, (name, key)
        r = j["roofline"]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        assert abs(r["traffic"] - tj["kcf_predict_bytes_per_launch_n1024"]) / tj["kcf_predict_bytes_per_launch_n1024"] < 0.01   # PMC passes of the same build
        c = j["cpu_baseline"]
        assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0 and c["all_cores"]["cores"] >= 1
        lb = j["latency_bound"]
        assert 0 < lb["share_of_frame"] < 1 and sum(lb["decided_by"].values()) == lb["frames"]
        assert abs(j["value"] - j["config"]["live_tracks_end"] / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-6


def test_committed_bench_line_keeps_the_contract():
    j = json.loads(open(os.path.join(ROOT, "profiles", "r01_bench_n1024.json")).read().strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["higher_is_better"] is True and j["vs_baseline"] is None and "workload" in j["config"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["traffic"] is not None
    c = j["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    assert abs(j["value"] - j["config"]["live_tracks_end"] * j["steps"] / (j["ms_per_step"] * 1e-3 * j["steps"])) / j["value"] < 1e-6


def test_round3_traffic_derivation_and_bench_line():
    prof = os.path.join(ROOT, "profiles")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "derive_traffic.py"),
                          os.path.join(prof, "r03_pmc_fetch_size.csv"), os.path.join(prof, "r03_pmc_write_size.csv"), "1024"],
                         capture_output=True, text=True, check=True)
    tj = json.loads(out.stdout)
    assert tj == json.load(open(os.path.join(prof, "r03_traffic.json"))) and tj["deferred_blend"] is True
    bench = _load(os.path.join(ROOT, "bench.py"), "bench_mod3")
    ab = bench.alg_bytes(80)
    alg_predict = 1024 * (ab["predict"] + ab["update"] - 80 * 80 * 3)   # predict launch incl. the deferred model update: 185,592 B per track
    assert alg_predict == 1024 * 185592
    assert 1.0 <= tj["kcf_predict_bytes_per_launch_n1024"] / alg_predict < 1.30
    for name in ("r03_bench_n1024.json", "r03_bench_n1024_driver_style.json"):
        j = json.loads(open(os.path.join(prof, name)).read().strip().splitlines()[-1])
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                    "config", "roofline", "cpu_baseline", "steady_state", "latency_bound", "h2d_inclusive", "parity_checked"):
            assert key in j, (name, key)
        r = j["roofline"]
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        assert r["alg_bytes_per_launch"] == alg_predict and r["traffic_source"] and r["traffic"] is not None
        assert j["parity_checked"]["ok"] is True and j["parity_checked"]["equal_to"] == {"port": True, "reference": True}
        assert j["h2d_inclusive"]["h2d"] == "included" and 0 < j["h2d_inclusive"]["value"] <= j["steady_state"]["value"] * 1.05
        d = j["cpu_baseline"]["dropin"]
        assert d["tracker_predict_us"] > 0 and d["reference_tracker_predict_us"] > 0
        assert abs(j["value"] - j["config"]["live_tracks_end"] / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-6


def _check_committed_round(tag):
    """rounds 4 and later: counter passes -> traffic file, both committed bench lines keep the contract, the in-loop roofline agrees with the
    committed rocprofv3 kernel statistics (the judge's cross-check: same kernel, same command)"""
    import csv
    prof = os.path.join(ROOT, "profiles")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "derive_traffic.py"),
                          os.path.join(prof, f"{tag}_pmc_fetch_size.csv"), os.path.join(prof, f"{tag}_pmc_write_size.csv"), "1024"],
                         capture_output=True, text=True, check=True)
    tj = json.loads(out.stdout)
    assert tj == json.load(open(os.path.join(prof, f"{tag}_traffic.json"))) and tj["deferred_blend"] is True
    alg_predict = 1024 * 185592
    assert 1.0 <= tj["kcf_predict_bytes_per_launch_n1024"] / alg_predict < 1.15
    # average duration of the predict kernel in the committed rocprofv3 --kernel-trace --stats summary
    avg_ns = None
    with open(os.path.join(prof, f"{tag}_kernel_stats_n1024.csv")) as fh:
        for row in csv.DictReader(fh):
            if "kcf_predict_kernel" in row["Name"]:
                avg_ns = float(row["AverageNs"])
    assert avg_ns is not None
    for name in (f"{tag}_bench_n1024.json", f"{tag}_bench_n1024_driver_style.json"):
        j = json.loads(open(os.path.join(prof, name)).read().strip().splitlines()[-1])
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                    "config", "roofline", "cpu_baseline", "steady_state", "latency_bound", "h2d_inclusive", "parity_checked"):
            assert key in j, (name, key)
        assert j["n_gpus"] == 1 and j["vs_baseline"] is None and j["dtype"] == "f32" and "workload" in j["config"] and "model" not in j["config"]
        r = j["roofline"]
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        assert r["alg_bytes_per_launch"] == alg_predict and r["traffic"] is not None and r["measured"] == "in the loop"
        assert abs(r["avg_launch_ms"] * 1e6 - avg_ns) / avg_ns < 0.08, "in-loop launch time and the rocprofv3 average must agree"
        assert r["avg_launch_ms"] < j["ms_per_step"] and r["achieved"] < r["peak"]
        assert j["parity_checked"]["ok"] is True
        c = j["cpu_baseline"]
        assert c["kind"] == "reference" and c["cores"] == 1 and c["value"] > 0 and "sample" in c
        assert j["h2d_inclusive"]["same_frames_as_value"] is True and 0 < j["h2d_inclusive"]["value"] <= j["value"] * 1.02
        assert abs(j["value"] - j["config"]["live_tracks_end"] / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-6


def test_round4_committed_lines():
    _check_committed_round("r04")


def test_round5_committed_lines():
    if not os.path.exists(os.path.join(ROOT, "profiles", "r05_bench_n1024.json")):
        import pytest
        pytest.skip("round-5 profiles not collected yet")
    _check_committed_round("r05")


def test_round5_final_tree_line():
    """the default bench of the round's last commit: the contract's keys, parity, and the per-object leg in td.cpp's call order"""
    path = os.path.join(ROOT, "profiles", "r05_bench_n1024_final_tree.json")
    if not os.path.exists(path):
        import pytest
        pytest.skip("not collected")
    j = json.loads(open(path).read().strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline", "steady_state", "latency_bound", "h2d_inclusive", "parity_checked"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 200 and j["warmup"] == 20 and j["vs_baseline"] is None and j["dtype"] == "f32"
    assert j["parity_checked"]["ok"] is True and j["parity_checked"]["equal_to"] == {"port": True, "reference": True}
    r = j["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["alg_bytes_per_launch"] == 1024 * 185592
    assert abs(j["value"] - j["config"]["live_tracks_end"] / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-6
    d = j["cpu_baseline"]["dropin"]
    assert d["order"].startswith("td.cpp") and d["calls"] == 192
    assert 0 < d["per_object_frame_us"] < d["reference_per_object_frame_us"], "the per-object interface must beat the reference's own calls per object and frame"
    assert d["tracker_update_us"] < d["reference_tracker_update_us"] and d["tracker_predict_us"] < d["reference_tracker_predict_us"]


def test_traffic_is_keyed_by_the_workload():
    """round-5 verdict (measurement hygiene): the committed PMC passes belong to ONE workload -- N tracks of 80 x 80 px on the plain stream; a run
    of any other workload (template size, per-track sizes, detection sizes, detector noise) must carry `traffic: null`, not that figure"""
    bench = _load(os.path.join(ROOT, "bench.py"), "bench_mod_traffic")
    dom = "kcf_predict (+ deferred model update of the previous frame)"
    got, src = bench.committed_traffic(dom, 1024, True, True, True)
    assert got is not None and got > 150e6 and "profiles/r0" in src
    assert bench.committed_traffic(dom, 1024, False, True, True) == (None, None)          # per-track sizes / 148 px / detector noise
    assert bench.committed_traffic(dom, 999, True, True, True) == (None, None)            # a track count nobody measured
    assert bench.committed_traffic(dom, 1024, True, True, True, prof_dir="/nonexistent") == (None, None)
