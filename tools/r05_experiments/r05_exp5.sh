#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05/e5; mkdir -p $O
MOT_KCF_K80=0 timeout 300 python tools/state_dump.py 48 128 8 4 5 21 --npz $O/state_k0.npz > $O/state_k0.txt 2>&1
MOT_KCF_K80=15 timeout 300 python tools/state_dump.py 48 128 8 4 5 21 --npz $O/state_k15.npz > $O/state_k15.txt 2>&1
python tools/state_diff.py $O/state_k0.npz $O/state_k15.npz > $O/state_diff_k0_k15.txt 2>&1; head -12 $O/state_diff_k0_k15.txt | cut -c1-1500
rm -f $O/state_k*.npz
# occupancy experiment: two workgroups per CU (default) against one
timeout 300 python tools/kcf_ablate.py --reps 6 --only-base 2>&1 | grep -v amdgpu | tee $O/ablate_two_per_cu.log
MOT_KCF_ONE_PER_CU=1 timeout 300 python tools/kcf_ablate.py --reps 6 --only-base 2>&1 | grep -v amdgpu | tee $O/ablate_one_per_cu.log
# the sparse-check soak WITHOUT snapshots on the build with the deterministic lifecycle lists
for e in "MOT_X=0" "MOT_SIDE_RESERVE=0" "MOT_LAP_TWO_BLOCK=0" "MOT_MID_IN_LAUNCH=0" "MOT_LAP_DENSE=0" "MOT_JOINED_LAUNCH=0"; do
  env $e timeout 700 python tools/lookahead_soak.py 48 8 5 4000 --sparse-checks --hammer --dump $O 2>&1 | grep -v amdgpu.ids | cut -c1-700 >> $O/soak_sparse.log
done
cat $O/soak_sparse.log
