#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
# Collects what DESIGN.md section 5 and profiles/ quote for one round:
#   gpurun_out/<tag>/bench.json             default bench.py run (C3)
#   gpurun_out/<tag>/kt/.../kernel_stats    rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/<tag>/pmc/<GROUP>/           one rocprofv3 --pmc pass per counter group (never mixed with a trace)
# Afterwards, in the development container:
#   cp gpurun_out/<tag>/kt/*/*kernel_stats.csv profiles/<tag>_kernel_stats.csv
#   cp gpurun_out/<tag>/bench.json profiles/<tag>_bench.json
#   python tools/pmc_summary.py gpurun_out/<tag>/pmc <tag>
set -u
tag=${1:?tag}
export TMPDIR=/tmp
out=gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out/pmc"
python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
tail -1 "$out/bench.json" | cut -c1-400
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$out/kt.log" 2>&1
for g in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $g --output-format csv -d "$out/pmc/$g" -- python3 bench.py --sites 131072 --steps 1 --warmup 0 --no-cpu-baseline > "$out/pmc_$g.log" 2>&1
done
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc/GRBM" -- python3 bench.py --sites 131072 --steps 1 --warmup 0 --no-cpu-baseline > "$out/pmc_GRBM.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$out/pmc/SQ" -- python3 bench.py --sites 131072 --steps 1 --warmup 0 --no-cpu-baseline > "$out/pmc_SQ.log" 2>&1
find "$out" -name "*kernel_trace.csv" -delete      # large; the stats summary is what is kept
ls "$out" "$out"/kt/* | head -20
