set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r4g
timeout -k 10 600 python -m pytest tests/test_gpu_fused.py -m gpu -x -q > gpurun_out/r4g/pytest.log 2>&1 || { tail -40 gpurun_out/r4g/pytest.log; exit 1; }
tail -3 gpurun_out/r4g/pytest.log
Q="--no-cpu-baseline --no-extra --no-pack-rate"
for wl in ${WLS:-fixedq c5}; do
  timeout -k 10 300 python bench.py --workload $wl --steps 3 --warmup 1 $Q > gpurun_out/r4g/bench_$wl.json 2> gpurun_out/r4g/bench_$wl.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4g/bench_$wl.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$wl", "%.3e"%d["value"], {k:round(r["kernel_ms_total"][k]/max(r["launches"][k],1),3) for k in r["kernel_ms_total"]}, d["ctx"]["fused"], d["ctx"]["fused_split"])
PY
done
