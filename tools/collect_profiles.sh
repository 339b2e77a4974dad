# evidence run of a round (GPU box; ROUND=r06 by default): bench lines, rocprofv3 kernel statistics, PMC passes, association probes.
# usage: [ROUND=r06] bash tools/collect_profiles.sh     (writes gpurun_out/$ROUND/final/; tools/install_profiles.sh copies the summaries into profiles/)
# Every rocprofv3 run gets a directory of its own, so each holds exactly one result set (no "newest file" guessing).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ROUND=${ROUND:-r06}
O=gpurun_out/$ROUND/final; rm -rf $O; mkdir -p $O
python bench.py > $O/bench_n1024.json 2> $O/bench_n1024.err
python bench.py --steps 20 --warmup 5 > $O/bench_n1024_driver.json 2>/dev/null
python bench.py --tracks 64 --no-cpu-baseline > $O/bench_n64.json 2>/dev/null
python bench.py --tracks 256 --no-cpu-baseline > $O/bench_n256.json 2>/dev/null
python bench.py --tracks 512 --no-cpu-baseline > $O/bench_n512.json 2>/dev/null
python bench.py --tracks 256 --size 148 --det-sizes 120 180 --no-cpu-baseline > $O/bench_n256_s148_multiscale.json 2>/dev/null
python bench.py --tracks 256 --size 150 --det-sizes 120 180 --per-track-sizes --no-cpu-baseline --h2d 0 > $O/bench_n256_per_track_sizes_120_180.json 2>/dev/null
python bench.py --tracks 1024 --det-sizes 64 96 --per-track-sizes --no-cpu-baseline --h2d 0 > $O/bench_n1024_per_track_sizes_64_96.json 2>/dev/null
python bench.py --tracks 256 --miss-pct 4 --fp-pct 3 --nms --steps 100 --warmup 10 --no-cpu-baseline --h2d 0 > $O/bench_n256_detector_noise.json 2>/dev/null
python bench.py --tracks 1000 --miss-pct 4 --fp-pct 3 --nms --steps 20 --warmup 5 --steady 0 --profile-frames 10 --no-cpu-baseline --h2d 0 > $O/bench_n1000_detector_noise.json 2>/dev/null
MOT_MK_LAZY=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --h2d 0 > $O/bench_n1024_driver_full_reset.json 2>/dev/null
MOT_MK_LAZY=0 python bench.py --no-cpu-baseline --h2d 0 > $O/bench_n1024_full_reset.json 2>/dev/null
for sz in 164 168 200; do python bench.py --tracks 64 --size $sz --steps 40 --warmup 10 --steady 0 --h2d 0 --no-cpu-baseline --profile-frames 10 > $O/bench_n64_s$sz.json 2>/dev/null; done
python tools/kcf_probe.py --frames 8 > $O/kcf_probe_n1024.log 2>&1
MOT_DBG_EXTRA=1 python tools/kcf_probe.py --frames 8 --tracks 256 --size 148 --det-sizes 120 180 > $O/kcf_probe_n256_s148.log 2>&1
python tools/assoc_probe.py 1024 30 > $O/assoc_probe_n1024.log 2>&1
python tools/assoc_trace.py 1024 23 8 > $O/assoc_trace_n1024_frame23.log 2>&1
python tools/assoc_trace.py 1024 7 8 > $O/assoc_trace_n1024_frame7.log 2>&1
./tools/ubench_latency > $O/ubench_latency.log 2>&1
# round 6: provisional commits on / off (same stream, unsynchronised loop), the host-fed loop behind a resident context, a timeline of tie frames
python tools/prov_probe.py 1024 221 20 > $O/prov_probe_n1024.log 2>&1; MOT_PROV=0 python tools/prov_probe.py 1024 221 20 > $O/prov_probe_n1024_off.log 2>&1
python tools/prov_probe.py 1024 26 5 > $O/prov_probe_driver.log 2>&1; MOT_PROV=0 python tools/prov_probe.py 1024 26 5 > $O/prov_probe_driver_off.log 2>&1
(python tools/hostfed_probe.py 1024 221 20 1; MOT_PROV=0 python tools/hostfed_probe.py 1024 221 20 1; python tools/hostfed_probe.py 1024 26 5 1) > $O/hostfed_probe.log 2>&1
MOT_PROV=0 python bench.py --no-cpu-baseline > $O/bench_n1024_prov_off.json 2>/dev/null; MOT_PROV=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_n1024_driver_prov_off.json 2>/dev/null
rocprofv3 --kernel-trace --output-format csv -d $O/ktrace_prov -- python3 tools/prov_probe.py 1024 60 5 > /dev/null 2>&1; python tools/trace_timeline.py $O/ktrace_prov/* 0.7 80 > $O/timeline_prov.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_default -- python3 bench.py --no-cpu-baseline --h2d 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_driver -- python3 bench.py --steps 20 --warmup 5 --steady 0 --h2d 0 --profile-frames 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_s148 -- python3 bench.py --tracks 256 --size 148 --det-sizes 120 180 --no-cpu-baseline --h2d 0 > /dev/null 2>&1
D="--steps 20 --warmup 5 --steady 0 --h2d 0 --no-cpu-baseline --profile-frames 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py $D > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py $D > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -- python3 bench.py $D > $O/pmc_sq.log 2>&1
S="--tracks 256 --size 148 --det-sizes 120 180 --no-cpu-baseline --h2d 0 --steps 20 --warmup 5 --steady 0 --profile-frames 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_s148 -- python3 bench.py $S > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_s148 -- python3 bench.py $S > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq_s148 -- python3 bench.py $S > /dev/null 2>&1
ls $O | head -60
