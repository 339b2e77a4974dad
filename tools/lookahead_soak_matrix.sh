#!/bin/bash
# The soak matrix of the look-ahead entry point (GPU box): track counts x scheduling variants x {state compare, sparse checks}.
# usage: tools/lookahead_soak_matrix.sh OUTDIR [SCALE]   (SCALE multiplies the repetition counts; 1 = about 25 minutes)
cd "$(dirname "$0")/.." || exit 1
O=${1:-gpurun_out/soak}; K=${2:-1}; mkdir -p $O
run() { # env-string, args...
  local e="$1"; shift
  echo "## env [$e] args [$*]" >> $O/soak_matrix.log
  env $e timeout 900 python tools/lookahead_soak.py "$@" --dump $O 2>&1 | grep -v amdgpu.ids >> $O/soak_matrix.log
}
for e in "MOT_X=0" "MOT_JOINED_LAUNCH=0" "MOT_SIDE_RESERVE=0" "MOT_LAP_TWO_BLOCK=0" "MOT_LAP_DENSE=0" "MOT_LAP_DENSE=1" "MOT_KCF_K80=0" "MOT_KCF_K80=15" "MOT_LOOKAHEAD=0"; do
  run "$e" 48 8 5 $((1200 * K)) --state --hammer --dirty
  run "$e" 48 8 5 $((1500 * K)) --sparse-checks --hammer
  run "$e" 300 6 4 $((250 * K)) --state --state-stride 3 --hammer
  run "$e" 1024 0 0 $((60 * K)) --state --state-stride 16 --hammer
done
run "MOT_X=0" 1024 0 0 $((400 * K)) --sparse-checks --hammer
run "MOT_X=0" 300 6 4 $((1000 * K)) --sparse-checks --hammer --dirty
grep -c '"mismatches": 0' $O/soak_matrix.log; grep -v '"mismatches": 0' $O/soak_matrix.log | grep mismatches | cut -c1-1500
