# k_gl2 validation (GPU box): the parity / fused / fuzz / scale suites with k_gl2 forced on wherever it can run (hooks build, VGL_GL2X=2), then with a
# tiny overflow pool (every workgroup with more than 2 three-base evaluations goes through k_gl_redo), then the shipped library's own choice; timings
set -e
export TMPDIR=/tmp
out=gpurun_out/gl2x; mkdir -p $out
H=$PWD/vcfgl_amd/lib/libvcfgl_hip_hooks.so
VGL_LIB=$H VGL_GL2X=2 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_scale_oracle.py tests/test_gpu_golden.py -m gpu -x -q > $out/pytest_on.log 2>&1 || { tail -30 $out/pytest_on.log; exit 1; }
tail -1 $out/pytest_on.log
VGL_LIB=$H VGL_GL2X=2 VGL_DEBUG_GL2_OVC=2 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > $out/pytest_redo.log 2>&1 || { tail -30 $out/pytest_redo.log; exit 1; }
tail -1 $out/pytest_redo.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_scale_oracle.py tests/test_gpu_fused.py -m gpu -x -q > $out/pytest_auto.log 2>&1 || { tail -30 $out/pytest_auto.log; exit 1; }
tail -1 $out/pytest_auto.log
Q="--no-cpu-baseline --no-extra --no-pack-rate"
for wl in ${*:-fixedq c4 c3 c2}; do
  timeout -k 10 300 python bench.py --workload $wl --steps 3 --warmup 1 $Q > $out/bench_$wl.json 2> $out/bench_$wl.err
  python - <<PY
import json
d=json.loads(open("$out/bench_$wl.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$wl", "%.3e"%d["value"], r["kernel_ms_per_launch"], "gl_wpb", d["ctx"]["gl_wpb"])
PY
done
