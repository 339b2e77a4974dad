#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05/e10; mkdir -p $O
for e in "MOT_X=0" "MOT_X=1"; do
  env $e timeout 800 python tools/lookahead_soak.py 1024 0 0 3000 --sparse-checks --hammer --frames 3 --trace --dump $O 2>&1 | grep -v amdgpu.ids >> $O/soak_short_trace.log
done
python - <<'PY'
import json
for ln in open("gpurun_out/r05/e10/soak_short_trace.log"):
    if not ln.startswith("{"): print(ln[:300]); continue
    j=json.loads(ln); print({k:j[k] for k in ("env","reps","mismatches","seconds")})
    for d in j["detail"]:
        print("  rep",d["rep"],"frame",d["frame"],"n_diff",d["n_diff"],d["first"][:3])
        for t in d.get("trace_diff",[]): print("     ",t)
PY
