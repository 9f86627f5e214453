#!/bin/bash
# usage (GPU box): tools/ab_time.sh <lib name> ...   -> kernel times of the bench workloads for each vcfgl_amd/lib_ab/<name>.so, interleaved twice
for rep in 1 2; do for lib in "$@"; do for w in ${AB_WORKLOADS:-c3 c5 fixedq gl1q precise}; do
VGL_LIB=$PWD/vcfgl_amd/lib_ab/$lib.so python bench.py --workload $w --sites 262144 --steps 3 --no-cpu-baseline --no-extra --no-pack-rate 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib $w rep$rep', '%.3e'%d['value'], d['roofline']['kernel_ms_per_launch'])"
done; done; done
