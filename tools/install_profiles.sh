#!/bin/bash
# copies the newest outputs of tools/collect_profiles.sh (gpurun_out/r03) into profiles/ under their committed names
set -e
cd "$(dirname "$0")/.."
R=gpurun_out/r03; newest() { ls -t $1 | head -1; }
cp $R/bench_n1024.json profiles/r03_bench_n1024.json; cp $R/bench_n1024_driver.json profiles/r03_bench_n1024_driver_style.json
for n in 64 256 512; do cp $R/bench_n$n.json profiles/r03_bench_n$n.json; done
for f in bench_n256_s148_multiscale bench_n256_per_track_sizes_120_180 bench_n1024_per_track_sizes_64_96 bench_n256_detector_noise bench_n1000_detector_noise bench_n1024_no_lookahead bench_n1024_no_deferred_blend bench_n1024_fused_update; do cp $R/$f.json profiles/r03_$f.json; done
cp $R/kcf_probe_n1024.log profiles/r03_kcf_probe_n1024.log
cp $(newest "$R/kstats_default/runc/*_kernel_stats.csv") profiles/r03_kernel_stats_n1024.csv
cp $(newest "$R/kstats_driver/runc/*_kernel_stats.csv") profiles/r03_kernel_stats_n1024_driver_style.csv
cp $(newest "$R/kstats_s148/runc/*_kernel_stats.csv") profiles/r03_kernel_stats_n256_s148_multiscale.csv
cp $(newest "$R/pmc_fetch/runc/*_counter_collection.csv") profiles/r03_pmc_fetch_size.csv
cp $(newest "$R/pmc_write/runc/*_counter_collection.csv") profiles/r03_pmc_write_size.csv
python tools/derive_traffic.py profiles/r03_pmc_fetch_size.csv profiles/r03_pmc_write_size.csv 1024 > profiles/r03_traffic.json
python tools/derive_sq.py $(newest "$R/pmc_sq/runc/*_counter_collection.csv") > profiles/r03_sq_counters.json
