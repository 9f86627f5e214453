# usage (GPU box): tools/kgl_ab.sh -> k_gl tuning overrides of the hooks build (workgroup size, sort, XCD map, run table) on fixed-q and C3
export VGL_LIB=$PWD/vcfgl_amd/lib/libvcfgl_hip_hooks.so
Q="--no-cpu-baseline --no-extra --no-pack-rate --steps 3 --warmup 1"
run() { python3 bench.py --workload $1 $Q 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 $2', '%.3e'%d['value'], {k:r['kernel_ms_per_launch'].get(k) for k in ('k_sample','k_gl')})"; }
for wl in fixedq c3; do
run $wl base
VGL_GL_WPB=4 run $wl wpb4
VGL_GL_SORT=0 run $wl nosort
VGL_XCD_MAP=0 run $wl noxcd
VGL_NO_GL2_RUN=1 run $wl norun
done
