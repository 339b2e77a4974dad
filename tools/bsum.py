#!/usr/bin/env python3
"""one-line summaries of bench.py output files"""
import json, sys
for f in sys.argv[1:]:
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    ss = j.get('steady_state') or {}
    print(f"{f}: {round(j['value'])} upd/s {j['ms_per_step']:.4f} ms | steady {round(ss.get('value', 0))}", end='')
    if 'h2d_inclusive' in j: print(f" | h2d {round(j['h2d_inclusive']['value'])}", end='')
    print()
    if 'kernel_ms' in j:
        print('   kernel_ms', {k[:12]: round(v, 4) for k, v in j['kernel_ms'].items()})
        lb = j['latency_bound']; print('   assoc mean/p50/p90/max', round(lb['ms_mean'], 4), round(lb['ms_p50'], 4), round(lb['ms_p90'], 4), round(lb['ms_max'], 4), lb['decided_by'], 'share', round(lb['share_of_frame'], 3))
        print('   roofline', j['roofline']['kernel'][:34], 'frac', round(j['roofline']['frac'], 4), 'ms', round(j['roofline']['avg_launch_ms'], 4))
