#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05/e12; mkdir -p $O
echo "== the regression test on a library built BEFORE the fix (expected: fails)" | tee $O/regress.log
MOT_AMD_LIB=$PWD/multiple-object-tracking_amd/libmot_amd_h0g0.so timeout 600 python -m pytest tests/test_gpu_devloop.py -q -k setup_fills 2>&1 | tail -12 | cut -c1-300 | tee -a $O/regress.log
echo "== the regression test on the fixed library (expected: passes)" | tee -a $O/regress.log
timeout 600 python -m pytest tests/test_gpu_devloop.py -q -k setup_fills 2>&1 | tail -3 | tee -a $O/regress.log
for v in "1024 0 0 20000 --frames 3" "48 8 5 10000"; do
  timeout 900 python tools/lookahead_soak.py $v --sparse-checks --hammer --dump $O 2>&1 | grep -v amdgpu.ids | cut -c1-700 >> $O/soak_fixed.log
done
cat $O/soak_fixed.log
