#!/bin/bash
# usage (development container, after tools/profile_round.sh ran on the GPU box): tools/collect_profiles.sh <tag> [workload ...]
# copies the kernel-trace summaries into profiles/ and writes profiles/<tag>_<wl>_pmc_summary.json + profiles/pmc_traffic.json
set -eu
tag=${1:?tag}; shift
wls=${*:-c3 c5 c5u8 c4 fixedq c2 gl1q precise alltags qsi16}
for wl in $wls; do
    f=$(ls -t gpurun_out/$tag/$wl/kt/*/*kernel_stats.csv | head -1)       # the newest (an earlier call's files under the same tag stay beside it)
    cp "$f" profiles/${tag}_${wl}_kernel_stats.csv
    [ -f gpurun_out/$tag/$wl/kernel_timed_stats.csv ] && cp gpurun_out/$tag/$wl/kernel_timed_stats.csv profiles/${tag}_${wl}_kernel_timed_stats.csv
    [ -n "${KTONLY:-}" ] && continue
    if [ "$wl" = c2 ]; then n=10000; else n=65536; fi
    python tools/pmc_summary.py gpurun_out/$tag/$wl/pmc ${tag}_${wl} $wl $n > /dev/null
    echo "$wl: $(grep -c . profiles/${tag}_${wl}_kernel_stats.csv) kernel rows"
done
[ -f gpurun_out/$tag/bench.json ] && tail -2 gpurun_out/$tag/bench.json > profiles/${tag}_bench.json || true
