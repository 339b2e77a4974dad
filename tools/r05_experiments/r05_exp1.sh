#!/bin/bash
# round-5 experiment 1 (GPU box): poison runs of the GPU suite, the folded sparse-update variant, first soak
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05/e1; mkdir -p $O
export TMPDIR=/tmp
echo "== folded sparse update (MOT_KCF_K80=15), plain" > $O/folded.log
MOT_KCF_K80=15 timeout 600 python -m pytest tests/test_gpu_devloop.py -q -k "test_device_loop_vs_oracle" >> $O/folded.log 2>&1
echo "== folded sparse update, LDS poison NaN" >> $O/folded.log
MOT_KCF_K80=15 MOT_LDS_POISON=0xFFFFFFFF timeout 600 python -m pytest tests/test_gpu_devloop.py -q -k "test_device_loop_vs_oracle" >> $O/folded.log 2>&1
echo "== folded sparse update, LDS poison 0" >> $O/folded.log
MOT_KCF_K80=15 MOT_LDS_POISON=0 timeout 600 python -m pytest tests/test_gpu_devloop.py -q -k "test_device_loop_vs_oracle" >> $O/folded.log 2>&1
echo "== default kernels, LDS poison NaN + HBM poison FF" >> $O/folded.log
MOT_POISON=0xFF MOT_LDS_POISON=0xFFFFFFFF timeout 600 python -m pytest tests/test_gpu_devloop.py -q -k "test_device_loop_vs_oracle" >> $O/folded.log 2>&1
MOT_POISON=0xFF MOT_LDS_POISON=0xFFFFFFFF timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_poison_ff.log 2>&1
MOT_POISON=0x7F MOT_LDS_POISON=0x7F7F7F7F timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_poison_7f.log 2>&1
for v in "48 8 5 150 --dirty" "48 8 5 150 --dirty --hammer" "48 8 5 150 --sparse-checks --hammer" "300 6 4 60 --dirty --hammer" "1024 0 0 30 --hammer"; do
  timeout 600 python tools/lookahead_soak.py $v >> $O/soak.log 2>&1
done
tail -3 $O/*.log
