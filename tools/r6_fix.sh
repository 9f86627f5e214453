# usage (GPU box): bash tools/r6_fix.sh <lib ...>  -> parity cases of the fixed-score sampler and the fused kernel on the working tree, then A/B on fixedq / c5 / c5u8 / c2
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r06_fix
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_scale_oracle.py tests/test_gpu_fuzz.py tests/test_gpu_extreme_shapes.py -x -q -k "not precise" > gpurun_out/r06_fix/pytest.log 2>&1 || { tail -30 gpurun_out/r06_fix/pytest.log; exit 1; }
tail -2 gpurun_out/r06_fix/pytest.log
AB_WORKLOADS="${AB_WORKLOADS:-fixedq c5 c5u8 c2}" bash tools/ab_time.sh "$@" > gpurun_out/r06_fix/ab.txt 2>&1
cat gpurun_out/r06_fix/ab.txt | sed 's/k_site.: [0-9.e-]*, //'
