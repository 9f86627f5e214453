#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points on the C3 flags (page-locked buffers, two tiles in flight, and the
synchronous call on pageable buffers) -- the numbers DESIGN.md quotes beside the device-resident bench (bench.py reports
the same as extra.host_path_c3)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
print(json.dumps(bench.host_path_rate(None, {"local_dev": 0}), indent=1))
