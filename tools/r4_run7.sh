set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r4h
timeout -k 10 600 python -m pytest tests/test_gpu_fused.py -m gpu -x -q > gpurun_out/r4h/pytest.log 2>&1 || { tail -40 gpurun_out/r4h/pytest.log; exit 1; }
tail -2 gpurun_out/r4h/pytest.log
SPECS='"c5 1000" "c5 2000" "c5 500"' 
export VGL_LIB=$PWD/vcfgl_amd/lib/libvcfgl_hip_hooks.so
Q="--no-cpu-baseline --no-extra --no-pack-rate --steps 3 --warmup 1"
for n in 500 1000 2000 4000; do sites=$(( 262144 * 1000 / n ))
  for nf in 0 1; do
    if [ $nf = 1 ]; then export VGL_NO_FUSE=1; else unset VGL_NO_FUSE; fi
    python3 bench.py --workload c5 --samples $n --sites $sites $Q 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('c5 N=$n', 'three kernels' if $nf else 'fused        ', '%.3e'%d['value'], {k:round(r['kernel_ms_total'][k]/max(r['launches'][k],1),3) for k in ('k_sample','k_gl')}, 'split', d['ctx']['fused_split'])"
  done
done
