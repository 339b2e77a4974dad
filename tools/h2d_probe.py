#!/usr/bin/env python3
"""H2D bandwidth of this box for the sizes the frame loop moves (2.76 MB frame, 24 KB detection list): pinned vs pageable,
one stream.  Informative; used to interpret bench.py's h2d_inclusive line."""
import time
import torch
for nbytes in (24 * 1024, 2764800, 16 * 2764800):
    dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    for name, host in (("pinned", torch.empty(nbytes, dtype=torch.uint8).pin_memory()), ("pageable", torch.empty(nbytes, dtype=torch.uint8))):
        for _ in range(3):
            dev.copy_(host, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 20
        for _ in range(n):
            dev.copy_(host, non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{nbytes:>10} B {name:9s}: {dt * 1e6:9.1f} us/copy  {nbytes / dt / 1e9:7.2f} GB/s")
