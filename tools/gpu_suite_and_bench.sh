# usage (GPU box): bash tools/gpu_suite_and_bench.sh <tag> [workload ...]  -> the whole -m gpu suite, then bench lines of the given workloads
set -e
export TMPDIR=/tmp
tag=${1:-r4x}; shift || true
mkdir -p gpurun_out/$tag
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest.log 2>&1 || { tail -40 gpurun_out/$tag/pytest.log; exit 1; }
tail -3 gpurun_out/$tag/pytest.log
Q="--no-cpu-baseline --no-extra --no-pack-rate"
for wl in ${*:-c3}; do
  timeout -k 10 300 python bench.py --workload $wl --steps 3 --warmup 1 $Q > gpurun_out/$tag/bench_$wl.json 2> gpurun_out/$tag/bench_$wl.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/$tag/bench_$wl.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$wl", "%.3e"%d["value"], r["kernel_ms_per_launch"])
PY
done
