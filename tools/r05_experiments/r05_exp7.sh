#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05/e7; mkdir -p $O
# does the direct update kernel (folded, inlined) + blend launch leave the same bits as the sparse kernel + deferred blend?
timeout 300 python tools/state_dump.py 48 128 8 4 6 21 --npz $O/s_default.npz > /dev/null 2>&1
MOT_DEFER_BLEND=0 timeout 300 python tools/state_dump.py 48 128 8 4 6 21 --npz $O/s_nodefer.npz > /dev/null 2>&1
MOT_SPLIT_UPDATE=0 timeout 300 python tools/state_dump.py 48 128 8 4 6 21 --npz $O/s_fused.npz > /dev/null 2>&1
echo "== default vs MOT_DEFER_BLEND=0"; python tools/state_diff.py $O/s_default.npz $O/s_nodefer.npz | grep -v "_flags" | tail -5 | cut -c1-300
echo "== default vs MOT_SPLIT_UPDATE=0"; python tools/state_diff.py $O/s_default.npz $O/s_fused.npz | grep -v "_flags" | tail -5 | cut -c1-300
rm -f $O/*.npz
# short runs, nothing synchronised, complete state compared at the end: every first update of every track is checked
for e in "MOT_X=0" "MOT_EVENT_SYSTEM=1" "MOT_X=1" "MOT_EVENT_SYSTEM=1"; do
  env $e timeout 900 python tools/lookahead_soak.py 1024 0 0 2500 --sparse-checks --state --hammer --frames 3 --dump $O 2>&1 | grep -v amdgpu.ids | cut -c1-1500 >> $O/soak_short.log
done
cat $O/soak_short.log
