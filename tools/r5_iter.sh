# one iteration on the float32 pool loop: bounds sweep, parity subset, C3 timing, instruction counts -- bash tools/r5_iter.sh <tag> [workloads]
set -e
export TMPDIR=/tmp
tag=${1:-r05_it}; shift || true
mkdir -p gpurun_out/$tag
timeout -k 10 600 python -m pytest tests/test_gpu_bounds.py -m gpu -x -q -s -k "float32_pool" > gpurun_out/$tag/bounds.log 2>&1 || { tail -30 gpurun_out/$tag/bounds.log; exit 1; }
grep "^pool32" gpurun_out/$tag/bounds.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale_oracle.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/$tag/pytest.log 2>&1 || { tail -40 gpurun_out/$tag/pytest.log; exit 1; }
tail -2 gpurun_out/$tag/pytest.log
Q="--no-cpu-baseline --no-extra --no-pack-rate"
for wl in ${*:-c3 qsi16}; do
  timeout -k 10 300 python bench.py --workload $wl --steps 3 --warmup 1 $Q > gpurun_out/$tag/bench_$wl.json 2> gpurun_out/$tag/bench_$wl.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/$tag/bench_$wl.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$wl", "%.3e"%d["value"], r["kernel_ms_per_launch"])
PY
done
bash tools/quick_pmc.sh c3 > gpurun_out/$tag/pmc_c3.txt 2>&1; grep "k_sample\|k_gl" gpurun_out/$tag/pmc_c3.txt
