# usage (GPU box): bash tools/r6_prec.sh <lib ...>  -> the --precise-gl 1 parity cases on the working tree's library, then A/B of the named libraries on `precise`
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r06_prec
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale_oracle.py tests/test_gpu_fuzz.py -x -q -k "precise or fuzz or deferred or extreme" > gpurun_out/r06_prec/pytest.log 2>&1 || { tail -30 gpurun_out/r06_prec/pytest.log; exit 1; }
tail -2 gpurun_out/r06_prec/pytest.log
AB_WORKLOADS="precise" bash tools/ab_time.sh "$@" > gpurun_out/r06_prec/ab.txt 2>&1
cat gpurun_out/r06_prec/ab.txt
