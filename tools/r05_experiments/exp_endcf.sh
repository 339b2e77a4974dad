#!/bin/bash
# Round 5: the folded sparse-update body (MOT_KCF_K80 bit 3) as hipcc builds it by default and with the inner EXEC restores kept
# (make -C multiple-object-tracking_amd/csrc endcf), against the general kernels: per-frame hashes of the whole tracker state.
mkdir -p gpurun_out; L=gpurun_out/r05_endcf.log; : > $L
P=multiple-object-tracking_amd
echo "== scan of the three libraries (tools/isa_exec0_scan.py)" >> $L
python tools/isa_exec0_scan.py $P/libmot_amd.so $P/libmot_amd_view.so $P/libmot_amd_view_endcf.so >> $L 2>&1
A="48 128 8 4 9 21"
echo "== general kernels (libmot_amd.so, MOT_KCF_K80=0): tools/state_dump.py $A" >> $L
MOT_KCF_K80=0 python tools/state_dump.py $A 2>/dev/null | grep '^frame ' > gpurun_out/h_general.txt; cat gpurun_out/h_general.txt >> $L
echo "== default build of the folded sparse update (libmot_amd_view.so, MOT_KCF_K80=15)" >> $L
MOT_AMD_LIB=$PWD/$P/libmot_amd_view.so MOT_KCF_K80=15 python tools/state_dump.py $A 2>/dev/null | grep '^frame ' > gpurun_out/h_view.txt; cat gpurun_out/h_view.txt >> $L
echo "== same source, -mllvm -amdgpu-remove-redundant-endcf=0 (libmot_amd_view_endcf.so, MOT_KCF_K80=15)" >> $L
MOT_AMD_LIB=$PWD/$P/libmot_amd_view_endcf.so MOT_KCF_K80=15 python tools/state_dump.py $A 2>/dev/null | grep '^frame ' > gpurun_out/h_endcf.txt; cat gpurun_out/h_endcf.txt >> $L
echo "== verdict" >> $L
cmp -s gpurun_out/h_general.txt gpurun_out/h_view.txt && echo "default build: EQUAL to the general kernels" >> $L || echo "default build: DIFFERS from the general kernels (frames: $(diff gpurun_out/h_general.txt gpurun_out/h_view.txt | grep -c '^>'))" >> $L
cmp -s gpurun_out/h_general.txt gpurun_out/h_endcf.txt && echo "endcf build:   EQUAL to the general kernels" >> $L || echo "endcf build:   DIFFERS from the general kernels" >> $L
tail -4 $L
