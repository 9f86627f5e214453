#!/bin/bash
# usage (GPU box): tools/kgl_probe.sh  -> k_gl at the C5 and C3 flags: time per launch for each VGL_GL_SORT, and SQ counters
export TMPDIR=/tmp
export VGL_LIB=$PWD/vcfgl_amd/lib/libvcfgl_hip_hooks.so   # VGL_GL_SORT is an override of the -DVGL_TEST_HOOKS build
out=gpurun_out/kglprobe; rm -rf $out; mkdir -p $out
Q="--no-cpu-baseline --no-extra --no-pack-rate"
for wl in c5 c3; do
  for gs in 2 0; do
    VGL_GL_SORT=$gs python3 bench.py --workload $wl --steps 3 --warmup 1 $Q 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$wl gl_sort $gs: %.4g evals/s, k_gl %.3f ms/launch, k_sample %.3f' % (d['value'], r['kernel_ms_per_launch']['k_gl'], r['kernel_ms_per_launch']['k_sample']))"
  done
done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/SQ -- python3 bench.py --workload c5 --sites 131072 --steps 1 --warmup 0 $Q > $out/log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/kglprobe/SQ/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if k.startswith("k_"): agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    print(k, {c: round(x / m["SQ_WAVES"], 1) for c, x in m.items() if c != "SQ_WAVES"})
PY
