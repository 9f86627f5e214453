#!/bin/bash
# usage (on the GPU box): tools/quick_traffic.sh -> HBM bytes per launch of every kernel of one bench tile (2*FETCH + WRITE, KiB -> bytes)
export TMPDIR=/tmp
rm -rf gpurun_out/qtr; mkdir -p gpurun_out/qtr
for g in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $g --output-format csv -d gpurun_out/qtr/$g -- python3 bench.py --sites 131072 --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-pack-rate > gpurun_out/qtr/$g.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(dict)
for g in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/qtr/{g}/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("k_"): agg[k].append(float(r["Counter_Value"]))
    for k, v in agg.items(): tot[k][g] = sum(v) / len(v)
for k, v in tot.items(): print(f"{k:28s} fetch {v['FETCH_SIZE']*1024/1e6:9.1f} MB (x2 corrected {2*v['FETCH_SIZE']*1024/1e6:9.1f})  write {v['WRITE_SIZE']*1024/1e6:9.1f} MB")
PY
