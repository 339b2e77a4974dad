#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05/e9; mkdir -p $O
for e in "MOT_X=0" "MOT_MK_LAZY=0" "MOT_MK_BATCH=0" "MOT_LAP_TWO_BLOCK=0"; do
  env $e timeout 300 python tools/assoc_soak.py 1024 150 --hammer 2>&1 | grep -v amdgpu.ids | cut -c1-1200 >> $O/assoc_soak.log
done
env MOT_X=0 timeout 300 python tools/assoc_soak.py 1024 100 --hammer --stream 0 2>&1 | grep -v amdgpu.ids | cut -c1-1200 >> $O/assoc_soak.log
cat $O/assoc_soak.log
