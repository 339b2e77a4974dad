#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05/e13; mkdir -p $O
echo "== the regression test on a library built BEFORE the fix (expected: fails)" | tee $O/regress.log
MOT_AMD_LIB=$PWD/multiple-object-tracking_amd/libmot_amd_h0g0.so timeout 300 python -m pytest tests/test_gpu_devloop.py -q -k setup_fills 2>&1 | tail -12 | cut -c1-400 | tee -a $O/regress.log
echo "== the regression test on the fixed library (expected: passes)" | tee -a $O/regress.log
timeout 300 python -m pytest tests/test_gpu_devloop.py -q -k setup_fills 2>&1 | tail -3 | tee -a $O/regress.log
./tools/memset_order_probe > $O/memset_order_probe.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_full.log 2>&1; tail -12 $O/pytest_full.log | cut -c1-300
