#!/bin/bash
# usage (GPU box): tools/traffic_by_n.sh  -> WRITE_SIZE / FETCH_SIZE of every kernel of one c3 tile for N = 1000 and N = 1024 samples:
# rows of 1000 samples are not a multiple of the 64-byte sectors the counters tally (4000 / 8000 bytes per row), rows of 1024 are
export TMPDIR=/tmp
for n in 1000 1024; do
  for g in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/trn/$n/$g; mkdir -p gpurun_out/trn/$n
    rocprofv3 --pmc $g --output-format csv -d gpurun_out/trn/$n/$g -- python3 bench.py --samples $n --sites 65536 --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-pack-rate > gpurun_out/trn/$n/$g.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections
for n in (1000, 1024):
    tot = collections.defaultdict(dict)
    for g in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"gpurun_out/trn/{n}/{g}/*/*counter_collection.csv")[0]
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            if k.startswith("k_"): agg[k].append(float(r["Counter_Value"]))
        for k, v in agg.items(): tot[k][g] = sum(v) / len(v)
    for k, v in sorted(tot.items()):
        e = 65536 * n
        print(f"N={n} {k:12s} fetch {v['FETCH_SIZE']*1024/e:7.2f} B/eval  write {v['WRITE_SIZE']*1024/e:7.2f} B/eval")
PY
