#!/usr/bin/env python3
"""Derive HBM bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only).

usage: derive_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <n_tracks> > profiles/rNN_traffic.json
(round 3 on: kcf_predict = predict + the deferred model update of the previous frame; kcf_features = the detection-feature launch)

Counters are in KiB.  On gfx950 FETCH_SIZE reports half of the streamed read bytes (MI355X_MICROARCH.md, HBM section),
so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  The median over the launches of a kernel is used (the first frame
spawns every track and is not steady state)."""
import csv, json, statistics, sys

def med(path, counter):
    per = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            for key in ("kcf_features", "kcf_predict", "kcf_update", "munkres", "assoc_min", "assoc_sub", "lap_rowscan", "lap_solve", "lap_verify", "mk_sparse", "mk_postcheck", "dl_lifecycle", "dl_scatter", "kalman"):
                if key in name:
                    per.setdefault(key, []).append(float(row["Counter_Value"]))
                    break
    # split update (device loop): every frame launches kcf_update twice, first the detection features, then the blend
    if "kcf_update" in per and "kcf_predict" in per and len(per["kcf_update"]) >= 2 * len(per["kcf_predict"]) - 2:
        u = per.pop("kcf_update")
        per["kcf_update_features"], per["kcf_update_blend"] = u[0::2], u[1::2]
    return {k: statistics.median(v) for k, v in per.items()}

def main():
    fetch, write, n = med(sys.argv[1], "FETCH_SIZE"), med(sys.argv[2], "WRITE_SIZE"), int(sys.argv[3])
    raw = {k: {"FETCH_SIZE": fetch.get(k), "WRITE_SIZE": write.get(k)} for k in sorted(set(fetch) | set(write))}
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), bench.py --steps 20 --warmup 5 "
                   f"--no-cpu-baseline --profile-frames 0, {n} tracks; median per launch, KiB. bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                   "(gfx950: FETCH_SIZE reports 1/2 of streamed read bytes, MI355X_MICROARCH.md)",
           "raw_kib": raw}
    if "kcf_features" in fetch:                                         # round 3: the feature-only launch has its own kernel and the blend rides in the predict
        out["deferred_blend"] = True
    for k in ("kcf_predict", "kcf_features", "kcf_update", "kcf_update_features", "kcf_update_blend"):
        if k in fetch and k in write:
            out[f"{k}_bytes_per_launch_n{n}"] = int((2 * fetch[k] + write[k]) * 1024)
    json.dump(out, sys.stdout, indent=1); print()

if __name__ == "__main__":
    main()
