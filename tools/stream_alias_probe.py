#!/usr/bin/env python3
"""Diagnostic: does the host path's copy / compute overlap survive other streams of the process?  (HIP maps streams onto a few
hardware queues; bench.py's host_path figure dropped from 53 to 35 GB/s when measured after the device-resident workloads.)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 0
big = int(sys.argv[2]) if len(sys.argv) > 2 else 0
streams = []
for i in range(n):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        x = torch.ones(1024, device="cuda") * 2
    streams.append(s)
torch.cuda.synchronize()
if big:                                   # allocate and free a lot of device memory through torch's allocator first
    t = [torch.empty((8 << 30,), dtype=torch.uint8, device="cuda") for _ in range(big)]
    del t
    torch.cuda.empty_cache()
r = bench.host_path_rate(None, {"local_dev": 0})
print(n, big, round(r["GBps_over_pcie"], 1), round(r["sync_pageable"]["GBps_over_pcie"], 1))
