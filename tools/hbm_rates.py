#!/usr/bin/env python3
"""Diagnostic: what the box's HBM delivers to simple streaming kernels -- write-only (fill), read-only (sum) and copy
(read + write) -- the denominators behind the HBM-bound figures of DESIGN.md section 5 (k_gl at depth 5 is write-dominated)."""
import torch
dev = torch.device("cuda", 0)
n = 1 << 30                                   # 4 GiB of float32
x = torch.empty(n, dtype=torch.float32, device=dev)
y = torch.empty(n, dtype=torch.float32, device=dev)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


gb = n * 4 / 1e9
t = timed(lambda: x.fill_(1.5)); print(f"fill  (write only): {gb / t:8.1f} GB/s written")
t = timed(lambda: x.sum());      print(f"sum   (read only) : {gb / t:8.1f} GB/s read")
t = timed(lambda: y.copy_(x));   print(f"copy  (read+write): {2 * gb / t:8.1f} GB/s moved ({gb / t:.1f} each way)")
