#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05/e8; mkdir -p $O
timeout 600 ./tools/event_scope_probe 200000 56 > $O/event_scope_probe.log 2>&1; cat $O/event_scope_probe.log
timeout 300 ./tools/event_scope_probe 100000 8 >> $O/event_scope_probe_8mb.log 2>&1; cat $O/event_scope_probe_8mb.log
timeout 1200 python -m pytest tests/test_gpu_devloop.py tests/test_gpu_nccl.py tests/test_gpu_parity.py -q -k "ranks or shard or rccl or bench" > $O/pytest_shard.log 2>&1; tail -8 $O/pytest_shard.log | cut -c1-400
