#!/usr/bin/env python3
"""Median SQ counters per launch and kernel from a rocprofv3 --pmc pass (counter_collection.csv).

usage: derive_sq.py <counter_collection.csv> > profiles/rNN_sq_counters.json"""
import csv, json, statistics, sys

KEYS = ("kcf_features", "kcf_predict", "kcf_update", "munkres", "assoc_sub", "lap_rowscan", "lap_solve", "lap_verify", "mk_sparse", "mk_postcheck", "kalman")
per = {}
with open(sys.argv[1], newline="") as f:
    for row in csv.DictReader(f):
        for k in KEYS:
            if k in row["Kernel_Name"]:
                per.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                break
out = {}
for k, cs in sorted(per.items()):
    o = {c: statistics.median(v) for c, v in sorted(cs.items())}
    if o.get("SQ_LDS_IDX_ACTIVE"):
        o["lds_bank_conflict_share"] = round(o.get("SQ_LDS_BANK_CONFLICT", 0.0) / o["SQ_LDS_IDX_ACTIVE"], 4)
    out[k] = o
print(json.dumps({"note": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY "
                          "SQ_ACTIVE_INST_ANY --kernel-trace (own pass), bench.py --steps 20 --warmup 5 --steady 0 --no-cpu-baseline --profile-frames 0, "
                          "1024 tracks; median per launch (kcf_update: feature and blend launches alternate, both in one list)",
                  "median_per_launch": out}, indent=1))
