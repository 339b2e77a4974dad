#!/bin/bash
# Round 5, last GPU minutes: the two-step column pass (libmot_amd_fftmix.so, -DMOT_FFT_MIXED=1) -- parity subset over generic template
# sizes, then per-track sizes 64-96 px and 200-px templates against the shipped library on the same box.
mkdir -p gpurun_out; L=gpurun_out/r05_fftmix.log; : > $L
export FM=$PWD/multiple-object-tracking_amd/libmot_amd_fftmix.so
echo "== parity subset with MOT_AMD_LIB=libmot_amd_fftmix.so" >> $L
MOT_AMD_LIB=$FM timeout 150 python -m pytest tests/test_gpu_parity.py tests/test_gpu_devloop.py -m gpu -q -x -k "kcf_sequence_golden or nonsquare or per_track_template_sizes or size_class" 2>&1 | tail -3 >> $L
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['ms_per_step'],4))"; }
for v in shipped fftmix; do
  [ $v = fftmix ] && export MOT_AMD_LIB=$FM || unset MOT_AMD_LIB
  echo "== $v: 1024 tracks, per-track sizes 64-96 px (value, ms/frame)" >> $L
  timeout 60 python bench.py --tracks 1024 --det-sizes 64 96 --per-track-sizes --no-cpu-baseline --h2d 0 --steps 60 --steady 0 --profile-frames 0 2>/dev/null | line >> $L
done
echo "== fftmix: 64 tracks, 200-px templates (shipped, this round's collection: 119 k, 0.538 ms)" >> $L
timeout 60 python bench.py --tracks 64 --size 200 --no-cpu-baseline --h2d 0 --steps 40 --warmup 10 --steady 0 --profile-frames 0 2>/dev/null | line >> $L
cat $L
