#!/bin/bash
# usage (on the GPU box): tools/quick_pmc_lds.sh [workload]  -> LDS / stall counters per kernel of one bench tile
export TMPDIR=/tmp
rm -rf gpurun_out/qpmc_lds; mkdir -p gpurun_out/qpmc_lds
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT --output-format csv -d gpurun_out/qpmc_lds/A -- python3 bench.py --workload ${1:-c3} --sites 131072 --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-pack-rate > gpurun_out/qpmc_lds/logA 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU --output-format csv -d gpurun_out/qpmc_lds/B -- python3 bench.py --workload ${1:-c3} --sites 131072 --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-pack-rate > gpurun_out/qpmc_lds/logB 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC --output-format csv -d gpurun_out/qpmc_lds/C -- python3 bench.py --workload ${1:-c3} --sites 131072 --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-pack-rate > gpurun_out/qpmc_lds/logC 2>&1
python3 - <<'PY'
import csv, glob, collections
for tag in "ABC":
    fs = glob.glob(f"gpurun_out/qpmc_lds/{tag}/*/*counter_collection.csv")
    if not fs:
        print(tag, "no output:", open(f"gpurun_out/qpmc_lds/log{tag}").read()[-600:]); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("k_sample") or k.startswith("k_gl"): agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        m = {c: sum(x) / len(x) for c, x in v.items()}
        w = m["SQ_WAVES"]
        print(tag, k[:34], {c: round(x / w, 1) for c, x in m.items() if c != "SQ_WAVES"})
PY
