#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05/e3; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log | cut -c1-300
for e in "MOT_X=0" "MOT_LAP_DENSE=0"; do
  env $e timeout 600 python tools/lookahead_soak.py 48 8 5 5000 --sparse-checks --hammer --snap --dump $O 2>&1 | grep -v amdgpu.ids >> $O/soak_snap.log
done
timeout 600 python tools/lookahead_soak.py 300 6 4 200 --state --state-stride 3 --hammer --dump $O 2>&1 | grep -v amdgpu.ids >> $O/soak_state.log
timeout 600 python tools/lookahead_soak.py 48 8 5 400 --state --hammer --dirty --dump $O 2>&1 | grep -v amdgpu.ids >> $O/soak_state.log
cut -c1-3000 $O/soak_snap.log; cut -c1-600 $O/soak_state.log
bash tools/r05_exp2.sh
