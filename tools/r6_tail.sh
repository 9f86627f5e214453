# usage (GPU box): bash tools/r6_tail.sh  -> the I16 cases of the parity / fuzz / scale / CLI suites after k_tail (tile-mode tail distances), then qsi16 / alltags timings
set -e
export TMPDIR=/tmp
out=gpurun_out/r06_tail; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_cli_fuzz.py tests/test_gpu_cli.py tests/test_gpu_golden.py -m gpu -x -q -k "i16 or I16 or tail or siteagg or strand or random or cli or golden or lean3 or optional" > $out/pytest_i16.log 2>&1 || { tail -40 $out/pytest_i16.log; exit 1; }
tail -3 $out/pytest_i16.log
timeout -k 10 600 python -m pytest tests/test_gpu_scale_oracle.py -m gpu -x -q -k "qsi16 or alltags" > $out/pytest_scale.log 2>&1 || { tail -40 $out/pytest_scale.log; exit 1; }
tail -3 $out/pytest_scale.log
Q="--no-cpu-baseline --no-extra --no-pack-rate"
for wl in qsi16 alltags; do
  timeout -k 10 300 python bench.py --workload $wl --steps 3 --warmup 1 $Q > $out/bench_$wl.json 2> $out/bench_$wl.err
  python - <<PY
import json
d=json.loads(open("$out/bench_$wl.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$wl", "%.3e"%d["value"], r["kernel_ms_per_launch"])
PY
done
