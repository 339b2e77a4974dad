#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05/e6; mkdir -p $O
MOT_KCF_K80=0 timeout 300 python tools/state_dump.py 48 128 8 4 5 21 --npz $O/state_k0.npz > $O/state_k0.txt 2>&1
MOT_AMD_LIB=$PWD/multiple-object-tracking_amd/libmot_amd_nosv.so MOT_KCF_K80=15 timeout 300 python tools/state_dump.py 48 128 8 4 5 21 --npz $O/state_k15_nosv.npz > $O/state_k15_nosv.txt 2>&1
MOT_AMD_LIB=$PWD/multiple-object-tracking_amd/libmot_amd_nosv.so MOT_KCF_K80=0 timeout 300 python tools/state_dump.py 48 128 8 4 5 21 --npz $O/state_k0_nosv.npz > $O/state_k0_nosv.txt 2>&1
echo "== K80=0 (product build) vs K80=15 (build with -amdgpu-spill-sgpr-to-vgpr=false)"; python tools/state_diff.py $O/state_k0.npz $O/state_k15_nosv.npz | tail -4 | cut -c1-400
echo "== K80=0 (product build) vs K80=0 (nosv build)"; python tools/state_diff.py $O/state_k0.npz $O/state_k0_nosv.npz | tail -2 | cut -c1-400
rm -f $O/*.npz
timeout 900 python -m pytest tests/test_gpu_variants.py tests/test_gpu_devloop.py -q -x -k "switch_combination or folded or finish_refuses or lookahead" > $O/pytest_new.log 2>&1; tail -6 $O/pytest_new.log | cut -c1-600
for e in "MOT_X=0" "MOT_LAP_TWO_BLOCK=0" "MOT_SIDE_RESERVE=0" "MOT_LAP_DENSE=0" "MOT_X=1" "MOT_JOINED_LAUNCH=0" "MOT_X=2" "MOT_LAP_TWO_BLOCK=0"; do
  env $e timeout 700 python tools/lookahead_soak.py 48 8 5 5000 --sparse-checks --hammer --trace --dump $O 2>&1 | grep -v amdgpu.ids >> $O/soak_trace.log
done
python - <<'PY'
import json
for ln in open("gpurun_out/r05/e6/soak_trace.log"):
    if not ln.startswith("{"): print(ln[:300]); continue
    j=json.loads(ln); print({k:j[k] for k in ("env","reps","mismatches","seconds")})
    for d in j["detail"]:
        print("  rep",d["rep"],"frame",d["frame"],d["first"])
        for t in d.get("trace_diff",[]): print("     ",t)
PY
