#!/bin/bash
# round-5 experiment 2 (GPU box): batched histogram -- parity, phase ablation of the predict launch, bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05/e2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x > $O/parity.log 2>&1; tail -3 $O/parity.log
timeout 600 python tools/kcf_ablate.py --reps 8 > $O/ablate.log 2>&1; cat $O/ablate.log | grep -v amdgpu
timeout 600 python bench.py --no-cpu-baseline --h2d 0 > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --h2d 0 > $O/bench_driver.json 2> $O/bench_driver.err; cut -c1-400 $O/bench_driver.json
timeout 300 python tools/kcf_probe.py --frames 8 > $O/kcf_probe.log 2>&1; tail -4 $O/kcf_probe.log
