# float32 pool loop: parity first, then timings -- bash tools/r5_f32.sh <tag>
set -e
export TMPDIR=/tmp
tag=${1:-r05_f32}
mkdir -p gpurun_out/$tag
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale_oracle.py tests/test_gpu_fuzz.py tests/test_gpu_scale.py tests/test_gpu_golden.py -m gpu -x -q > gpurun_out/$tag/pytest.log 2>&1 || { tail -40 gpurun_out/$tag/pytest.log; exit 1; }
tail -3 gpurun_out/$tag/pytest.log
Q="--no-cpu-baseline --no-extra --no-pack-rate"
for wl in c3 c4 qsi16 alltags gl1q; do
  timeout -k 10 300 python bench.py --workload $wl --steps 3 --warmup 1 $Q > gpurun_out/$tag/bench_$wl.json 2> gpurun_out/$tag/bench_$wl.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/$tag/bench_$wl.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$wl", "%.3e"%d["value"], r["kernel_ms_per_launch"])
PY
done
python tools/redo_rate.py > gpurun_out/$tag/redo_rate.txt 2>&1 || true
tail -5 gpurun_out/$tag/redo_rate.txt
