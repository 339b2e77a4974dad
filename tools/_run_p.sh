cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04/p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_devloop.py -m gpu -x -q -k "kcf_sequence_golden or fhog_golden or sharded_two_ranks" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for sz in 80 148 164 168 200 240; do python bench.py --tracks 64 --size $sz --steps 40 --warmup 10 --steady 0 --h2d 0 --no-cpu-baseline --profile-frames 10 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('size', $sz, 'value', round(j['value']), 'ms/step', round(j['ms_per_step'],3), 'roofline', j.get('roofline',{}).get('frac'), j.get('roofline',{}).get('avg_launch_ms'))"; done
S148="--tracks 256 --size 148 --det-sizes 120 180 --no-cpu-baseline --h2d 0"
python bench.py $S148 > $O/bench_n256_s148.json 2>/dev/null; cut -c1-200 $O/bench_n256_s148.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_s148 -- python3 bench.py $S148 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_s148 -- python3 bench.py $S148 --steps 20 --warmup 5 --steady 0 --profile-frames 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_s148 -- python3 bench.py $S148 --steps 20 --warmup 5 --steady 0 --profile-frames 0 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq_s148 -- python3 bench.py $S148 --steps 20 --warmup 5 --steady 0 --profile-frames 0 > /dev/null 2>&1
find $O -name "*kernel_stats.csv" | head; 
