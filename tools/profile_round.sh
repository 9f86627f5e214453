#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag> [workload ...]      (default: every bench workload)
# Collects what DESIGN.md section 5 and profiles/ quote for one round:
#   gpurun_out/<tag>/bench.json                   default bench.py run (C3 + the other configurations as `extra`)  [skipped with NOBENCH=1]
#   gpurun_out/<tag>/<wl>/kt/.../kernel_stats     rocprofv3 --kernel-trace --stats of the workload's bench command (1 warm-up + 3 timed steps)
#   gpurun_out/<tag>/<wl>/kernel_timed_stats.csv  the same trace, timed steps only (tools/trace_timed.py)      [KTONLY=1: stop here, no PMC passes]
#   gpurun_out/<tag>/<wl>/pmc/<GROUP>/            one rocprofv3 --pmc pass per counter group (never mixed with a trace)
# Afterwards, in the development container: tools/collect_profiles.sh <tag> [workload ...]
set -u
tag=${1:?tag}; shift
wls=${*:-c3 c5 c5u8 c4 fixedq c2 gl1q precise alltags qsi16}
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p "$out"
if [ -z "${NOBENCH:-}" ]; then
    python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
    tail -1 "$out/bench.json" | cut -c1-300
fi
Q="--no-cpu-baseline --no-extra --no-pack-rate"
for wl in $wls; do
    rm -rf "$out/$wl"; mkdir -p "$out/$wl/pmc"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$wl/kt" -- python3 bench.py --workload $wl --steps 3 --warmup 1 $Q > "$out/$wl/kt.log" 2>&1
    # the same trace restricted to the three timed steps (what bench.py's HIP events cover): tools/trace_timed.py
    for f in "$out/$wl"/kt/*/*kernel_trace.csv; do [ -f "$f" ] && python3 tools/trace_timed.py "$f" 1 3 > "$out/$wl/kernel_timed_stats.csv"; done
    [ -n "${KTONLY:-}" ] && { echo "$wl traced"; continue; }
    if [ "$wl" = c2 ]; then P="--workload $wl --steps 2 --warmup 0 $Q"; else P="--workload $wl --sites 131072 --steps 1 --warmup 0 $Q"; fi
    for g in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $g --output-format csv -d "$out/$wl/pmc/$g" -- python3 bench.py $P > "$out/$wl/pmc_$g.log" 2>&1
    done
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$out/$wl/pmc/SQ" -- python3 bench.py $P > "$out/$wl/pmc_SQ.log" 2>&1
    echo "$wl profiled"
done
find "$out" -name "*kernel_trace.csv" -delete      # large; the stats summary is what is kept
find "$out" -name "*.db" -delete
ls "$out"
