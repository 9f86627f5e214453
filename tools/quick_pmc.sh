#!/bin/bash
# usage (on the GPU box): tools/quick_pmc.sh [workload]  -> per-kernel VALU instruction counts of one bench tile
export TMPDIR=/tmp
rm -rf gpurun_out/qpmc; mkdir -p gpurun_out/qpmc
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/qpmc/SQ -- python3 bench.py --workload ${1:-c3} --sites 131072 --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-pack-rate > gpurun_out/qpmc/log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/qpmc/SQ/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if k.startswith("k_"): agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    w = m["SQ_WAVES"]
    print(f"{k:14s} VALU/wave {m['SQ_INSTS_VALU']/w:9.0f}  SALU/wave {m['SQ_INSTS_SALU']/w:8.0f}  lanes/VALU {m['SQ_THREAD_CYCLES_VALU']/m['SQ_INSTS_VALU']:5.1f}  "
          f"VALU-active cyc/wave {4*m['SQ_ACTIVE_INST_VALU']/w:9.0f}  wave cyc {4*m['SQ_WAVE_CYCLES']/w:9.0f}  wait_any {m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:.2f} wait_inst {m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES']:.2f}")
PY
