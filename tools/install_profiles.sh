#!/bin/bash
# copies the outputs of tools/collect_profiles.sh (gpurun_out/$ROUND/final, ROUND=r06 by default) into profiles/ under their committed names.  Every rocprofv3 run
# has a directory of its own holding exactly one result set: `one` fails loudly if that is ever not so.
set -e
cd "$(dirname "$0")/.."
# NOTE: gpurun MERGES into gpurun_out/: delete gpurun_out/$ROUND/final locally before a new collection, or `one` below trips over the previous run
ROUND=${ROUND:-r06}
R=gpurun_out/$ROUND/final
one() { local n; n=$(ls $1 2>/dev/null | wc -l); if [ "$n" != "1" ]; then echo "expected exactly one file for $1, found $n" >&2; exit 1; fi; ls $1; }
cp $R/bench_n1024.json profiles/${ROUND}_bench_n1024.json; cp $R/bench_n1024_driver.json profiles/${ROUND}_bench_n1024_driver_style.json
for n in 64 256 512; do cp $R/bench_n$n.json profiles/${ROUND}_bench_n$n.json; done
for f in bench_n256_s148_multiscale bench_n256_per_track_sizes_120_180 bench_n1024_per_track_sizes_64_96 bench_n256_detector_noise bench_n1000_detector_noise bench_n1024_driver_full_reset bench_n1024_full_reset bench_n64_s164 bench_n64_s168 bench_n64_s200; do cp $R/$f.json profiles/${ROUND}_$f.json; done
cp $R/kcf_probe_n1024.log profiles/${ROUND}_kcf_probe_n1024.log
cp $R/kcf_probe_n256_s148.log profiles/${ROUND}_kcf_probe_n256_s148.log
cp $R/assoc_probe_n1024.log profiles/${ROUND}_assoc_probe_n1024.log
cp $R/assoc_trace_n1024_frame23.log profiles/${ROUND}_assoc_trace_n1024_frame23.log; cp $R/assoc_trace_n1024_frame7.log profiles/${ROUND}_assoc_trace_n1024_frame7.log
cp $R/ubench_latency.log profiles/${ROUND}_ubench_latency.log
cp $(one "$R/kstats_default/*/*_kernel_stats.csv") profiles/${ROUND}_kernel_stats_n1024.csv
cp $(one "$R/kstats_driver/*/*_kernel_stats.csv") profiles/${ROUND}_kernel_stats_n1024_driver_style.csv
cp $(one "$R/kstats_s148/*/*_kernel_stats.csv") profiles/${ROUND}_kernel_stats_n256_s148_multiscale.csv
cp $(one "$R/pmc_fetch/*/*_counter_collection.csv") profiles/${ROUND}_pmc_fetch_size.csv
cp $(one "$R/pmc_write/*/*_counter_collection.csv") profiles/${ROUND}_pmc_write_size.csv
python tools/derive_traffic.py profiles/${ROUND}_pmc_fetch_size.csv profiles/${ROUND}_pmc_write_size.csv 1024 > profiles/${ROUND}_traffic.json
python tools/derive_sq.py $(one "$R/pmc_sq/*/*_counter_collection.csv") > profiles/${ROUND}_sq_counters.json
cp $(one "$R/pmc_fetch_s148/*/*_counter_collection.csv") profiles/${ROUND}_pmc_fetch_size_s148.csv
cp $(one "$R/pmc_write_s148/*/*_counter_collection.csv") profiles/${ROUND}_pmc_write_size_s148.csv
python tools/derive_traffic.py profiles/${ROUND}_pmc_fetch_size_s148.csv profiles/${ROUND}_pmc_write_size_s148.csv 256 > profiles/${ROUND}_traffic_s148.json
python tools/derive_sq.py $(one "$R/pmc_sq_s148/*/*_counter_collection.csv") > profiles/${ROUND}_sq_counters_s148.json
for f in prov_probe_n1024 prov_probe_n1024_off prov_probe_driver prov_probe_driver_off hostfed_probe timeline_prov bench_n1024_prov_off bench_n1024_driver_prov_off; do cp $R/$f.* profiles/${ROUND}_$(basename $R/$f.*); done
python tools/kernel_resources.py multiple-object-tracking_amd/libmot_amd.so > profiles/${ROUND}_kernel_resources.txt
