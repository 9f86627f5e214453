set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r4c
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bounds.py -m gpu -x -q -k "siteagg or quot or strand or alltags or i16" > gpurun_out/r4c/pytest.log 2>&1 || { tail -30 gpurun_out/r4c/pytest.log; exit 1; }
tail -3 gpurun_out/r4c/pytest.log
timeout -k 10 900 python -m pytest tests/test_gpu_scale_oracle.py -m gpu -x -q -k "alltags or adjbins or fixedq" > gpurun_out/r4c/pytest2.log 2>&1 || { tail -30 gpurun_out/r4c/pytest2.log; exit 1; }
tail -3 gpurun_out/r4c/pytest2.log
Q="--no-cpu-baseline --no-extra --no-pack-rate"
for wl in ${WLS:-alltags qsi16}; do
  timeout -k 10 300 python bench.py --workload $wl --steps 3 --warmup 1 $Q > gpurun_out/r4c/bench_$wl.json 2> gpurun_out/r4c/bench_$wl.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4c/bench_$wl.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$wl", "%.3e"%d["value"], {k:round(r["kernel_ms_total"][k]/max(r["launches"][k],1),3) for k in r["kernel_ms_total"]})
PY
done
