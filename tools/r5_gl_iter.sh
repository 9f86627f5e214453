# one iteration on k_gl: parity subsets, then fixed-q / C5 / C3 timing -- bash tools/r5_gl_iter.sh <tag> [workloads]
set -e
export TMPDIR=/tmp
tag=${1:-r05_gl}; shift || true
mkdir -p gpurun_out/$tag
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -m gpu -x -q -k "${GL_K:-gl2 or fused or parity}" > gpurun_out/$tag/pytest.log 2>&1 || { tail -40 gpurun_out/$tag/pytest.log; exit 1; }
tail -2 gpurun_out/$tag/pytest.log
Q="--no-cpu-baseline --no-extra --no-pack-rate"
for wl in ${*:-fixedq c5 c3}; do
  timeout -k 10 300 python bench.py --workload $wl --steps 3 --warmup 1 $Q > gpurun_out/$tag/bench_$wl.json 2> gpurun_out/$tag/bench_$wl.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/$tag/bench_$wl.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$wl", "%.3e"%d["value"], r["kernel_ms_per_launch"])
PY
done
