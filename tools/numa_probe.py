import os, sys, json, subprocess
print(subprocess.run("lscpu | grep -i -E 'numa|socket|model name' ; cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' '; echo; nproc", shell=True, capture_output=True, text=True).stdout)
ROOT="/root/repo"
for cores in ([0,1,2,3,4,5,6,7], list(range(64,72)), list(range(128,136)), list(range(192,200))):
    code = f"import os; os.sched_setaffinity(0, {set(cores)}); import sys; sys.path.insert(0,'{ROOT}'); sys.path.insert(0,'{ROOT}/tests'); import bench, json; r=bench.host_path_rate(None, {{'local_dev':0}}); print({cores[0]}, round(r['GBps_over_pcie'],1), round(r['sync_pageable']['GBps_over_pcie'],1))"
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=200)
        print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
    except Exception as e:
        print(cores[0], "err", e)
