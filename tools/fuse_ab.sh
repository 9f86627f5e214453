#!/bin/bash
# usage (GPU box): tools/fuse_ab.sh  -> fused kernel against the three-kernel path (VGL_NO_FUSE=1, hooks build) for fixed-score shapes:
# workload, samples per site; evaluations/s and kernel ms per launch
export VGL_LIB=$PWD/vcfgl_amd/lib/libvcfgl_hip_hooks.so
Q="--no-cpu-baseline --no-extra --no-pack-rate --steps 3 --warmup 1"
for spec in ${SPECS:-"fixedq 500" "fixedq 1000" "fixedq 2000" "c5 500" "c5 1000" "c5 2000"}; do
  set -- $spec; wl=$1; n=$2; sites=$(( 262144 * 1000 / n ))
  for nf in 0 1; do
    if [ $nf = 1 ]; then export VGL_NO_FUSE=1; else unset VGL_NO_FUSE; fi
    python3 bench.py --workload $wl --samples $n --sites $sites $Q 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$wl N=$n', 'three kernels' if $nf else 'fused        ', '%.3e'%d['value'], {k:r['kernel_ms_per_launch'].get(k) for k in ('k_depth','k_sample','k_gl')}, 'split', d['ctx']['fused_split'])"
  done
done
