#!/bin/bash
# usage (GPU box): AB_WORKLOADS="fixedq c5" tools/ab_env.sh "VAR=1" "VAR2=1 VAR3=2" ...  -> kernel times under the hooks library with each environment setting
# ("-" = no override), interleaved twice
export VGL_LIB=$PWD/vcfgl_amd/lib/libvcfgl_hip_hooks.so
for rep in 1 2; do for cfg in "$@"; do for w in ${AB_WORKLOADS:-fixedq c5}; do
[ "$cfg" = "-" ] && e="" || e="$cfg"
env $e python bench.py --workload $w --sites 262144 --steps 3 --no-cpu-baseline --no-extra --no-pack-rate 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$cfg] $w rep$rep', '%.3e'%d['value'], d['roofline']['kernel_ms_per_launch'])"
done; done; done
