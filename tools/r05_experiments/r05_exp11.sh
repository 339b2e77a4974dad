#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05/e11; mkdir -p $O
timeout 900 ./tools/event_scope_probe 100000 56 1 > $O/event_scope_probe_mode1.log 2>&1; cat $O/event_scope_probe_mode1.log
for e in "MOT_X=0" "MOT_EVENT_SYSTEM=1" "MOT_X=1" "MOT_EVENT_SYSTEM=1"; do
  env $e timeout 600 python tools/lookahead_soak.py 1024 0 0 15000 --sparse-checks --hammer --frames 3 --dump $O 2>&1 | grep -v amdgpu.ids | cut -c1-900 >> $O/soak_short_ab.log
done
cat $O/soak_short_ab.log
