#!/bin/bash
# per-object drop-in interface in td.cpp's call order: updates that return with their launch queued (default) against MOT_ZC_ASYNC=0
mkdir -p gpurun_out; L=gpurun_out/r05_zc_async.log; : > $L
for rep in 1 2; do for v in 1 0; do
  echo "== MOT_ZC_ASYNC=$v (run $rep)" >> $L
  MOT_ZC_ASYNC=$v python -c "
import bench, json
d = bench.dropin_timing(0)
print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in d.items() if k != 'note'}))" 2>/dev/null >> $L
done; done
cat $L
