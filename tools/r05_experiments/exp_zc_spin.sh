#!/bin/bash
# per-call latency of the drop-in interface: polling wait (default) against hipStreamSynchronize (MOT_ZC_SPIN=0), twice each
mkdir -p gpurun_out; L=gpurun_out/r05_zc_spin.log; : > $L
for rep in 1 2; do for v in 1 0; do
  echo "== MOT_ZC_SPIN=$v (run $rep)" >> $L
  MOT_ZC_SPIN=$v python -c "
import bench, json
d = bench.dropin_timing(0)
print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in d.items() if k != 'note'}))" 2>/dev/null >> $L
done; done
cat $L
