#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05/postfix; mkdir -p $O
for e in "MOT_X=0" "MOT_JOINED_LAUNCH=0" "MOT_SIDE_RESERVE=0" "MOT_LAP_TWO_BLOCK=0" "MOT_LAP_DENSE=0" "MOT_LOOKAHEAD=0"; do
  env $e timeout 400 python tools/lookahead_soak.py 48 8 5 5000 --sparse-checks --hammer --dump $O 2>&1 | grep -v amdgpu.ids | cut -c1-600 >> $O/soak_postfix.log
done
env MOT_X=0 timeout 600 python tools/lookahead_soak.py 1024 0 0 15000 --sparse-checks --hammer --frames 3 --dump $O 2>&1 | grep -v amdgpu.ids | cut -c1-600 >> $O/soak_postfix.log
env MOT_X=0 timeout 400 python tools/lookahead_soak.py 300 6 4 300 --state --state-stride 3 --hammer --dirty --dump $O 2>&1 | grep -v amdgpu.ids | cut -c1-600 >> $O/soak_postfix.log
timeout 300 python -m pytest tests/test_gpu_soak.py -q 2>&1 | tail -3 >> $O/soak_postfix.log
cat $O/soak_postfix.log
