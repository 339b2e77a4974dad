#!/bin/bash
# single template sizes, shipped library against libmot_amd_fftmix.so: 96 px (24 cells = 4 x 6), 64 px (16 = 4 x 4), 88 px (22 = 2 x 11)
mkdir -p gpurun_out; L=gpurun_out/r05_fftmix_sizes.log; : > $L
FM=$PWD/multiple-object-tracking_amd/libmot_amd_fftmix.so
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['ms_per_step'],4), d['kernel_ms'])"; }
for sz in 96 64; do for v in shipped fftmix; do
  [ $v = fftmix ] && export MOT_AMD_LIB=$FM || unset MOT_AMD_LIB
  echo "== $v: 512 tracks, $sz px" >> $L
  timeout 50 python bench.py --tracks 512 --size $sz --no-cpu-baseline --h2d 0 --steps 40 --warmup 10 --steady 0 --profile-frames 10 2>/dev/null | line >> $L
done; done
cat $L
