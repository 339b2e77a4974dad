#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05/final_pre; mkdir -p $O
echo "== regression test, library built BEFORE the fix (expected: fails)" | tee $O/regress.log
MOT_AMD_LIB=$PWD/multiple-object-tracking_amd/libmot_amd_h0g0.so timeout 300 python -m pytest tests/test_gpu_devloop.py -q -k setup_fills 2>&1 | tail -14 | cut -c1-300 | tee -a $O/regress.log
echo "== regression test, fixed library (expected: passes)" | tee -a $O/regress.log
timeout 900 python -m pytest tests/test_gpu_devloop.py -q -k "setup_fills or folded" 2>&1 | tail -4 | tee -a $O/regress.log
./tools/memset_order_probe > $O/memset_order_probe.log 2>&1; cat $O/memset_order_probe.log
ROUND=r05 bash tools/collect_profiles.sh > $O/collect.log 2>&1; tail -3 $O/collect.log
cp $O/memset_order_probe.log gpurun_out/r05/final/memset_order_probe.log; cp $O/regress.log gpurun_out/r05/final/regression_test_old_vs_fixed.log
head -c 600 gpurun_out/r05/final/bench_n1024.json; echo; head -c 400 gpurun_out/r05/final/bench_n1024_driver.json; echo
