#!/bin/bash
# usage (GPU box): tools/kgl_env_probe.sh [workload ...] -> k_gl per launch under the hooks library's VGL_GL_SORT / VGL_GL_WPB overrides
export VGL_LIB=$PWD/vcfgl_amd/lib/libvcfgl_hip_hooks.so
Q="--no-cpu-baseline --no-extra --no-pack-rate --sites 262144 --steps 3 --warmup 1"
for wl in ${*:-c3 fixedq}; do
  for cfg in "VGL_GL_SORT=2 VGL_GL_WPB=8" "VGL_GL_SORT=1 VGL_GL_WPB=8" "VGL_GL_SORT=0 VGL_GL_WPB=8" "VGL_GL_SORT=2 VGL_GL_WPB=4" "VGL_GL_SORT=0 VGL_GL_WPB=4"; do
    env $cfg python3 bench.py --workload $wl $Q 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']['kernel_ms_per_launch']
print('$wl $cfg: k_gl %.3f ms, k_sample %.3f, gl_sort %s wpb %s' % (r['k_gl'], r['k_sample'], d['ctx']['gl_sort'], d['ctx']['gl_wpb']))"
  done
done
