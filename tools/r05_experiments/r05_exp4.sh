#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05/e4; mkdir -p $O
for e in "MOT_X=0" "MOT_LAP_DENSE=0"; do
  env $e timeout 700 python tools/lookahead_soak.py 48 8 5 6000 --sparse-checks --hammer --snap --dump $O 2>&1 | grep -v amdgpu.ids >> $O/soak_snap.log
done
cut -c1-6000 $O/soak_snap.log
MOT_KCF_K80=0 timeout 300 python tools/state_dump.py 48 128 8 4 9 21 --npz $O/state_k0.npz > $O/state_k0.txt 2>&1
MOT_KCF_K80=15 timeout 300 python tools/state_dump.py 48 128 8 4 9 21 --npz $O/state_k15.npz > $O/state_k15.txt 2>&1
MOT_KCF_K80=7 timeout 300 python tools/state_dump.py 48 128 8 4 9 21 --npz $O/state_k7.npz > $O/state_k7.txt 2>&1
python tools/state_diff.py $O/state_k0.npz $O/state_k15.npz > $O/state_diff_k0_k15.txt 2>&1; head -40 $O/state_diff_k0_k15.txt
python tools/state_diff.py $O/state_k0.npz $O/state_k7.npz > $O/state_diff_k0_k7.txt 2>&1; tail -2 $O/state_diff_k0_k7.txt
rm -f $O/state_k*.npz
for rep in 1 2; do for v in h0g0 default h1g0 h1g1; do
  L=multiple-object-tracking_amd/libmot_amd_$v.so; [ $v = default ] && L=multiple-object-tracking_amd/libmot_amd.so
  MOT_AMD_LIB=$PWD/$L timeout 300 python bench.py --no-cpu-baseline --h2d 0 --steps 150 --warmup 20 --steady 0 --profile-frames 5 > $O/ab_${v}_$rep.json 2>/dev/null
  python - <<PY
import json; j=json.load(open("$O/ab_${v}_$rep.json")); print("$v rep $rep: value %.0f  in-loop predict %.1f us  isolated %.1f us" % (j["value"], j["roofline"]["avg_launch_ms"]*1e3, j["roofline"]["isolated"]["avg_launch_ms"]*1e3))
PY
done; done
