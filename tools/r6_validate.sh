# usage (GPU box): bash tools/r6_validate.sh  -> the whole -m gpu suite, a fuzz run with fresh seeds, the host-program fuzz (logs under gpurun_out/r06_val)
set -e
export TMPDIR=/tmp
out=gpurun_out/r06_val; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
VGL_FUZZ_CHUNKS=${FUZZ_CHUNKS:-100} VGL_FUZZ_SEED=${FUZZ_SEED:-616000} timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k random_configurations > $out/fuzz_seed${FUZZ_SEED:-616000}.log 2>&1 || { tail -40 $out/fuzz_seed${FUZZ_SEED:-616000}.log; exit 1; }
tail -2 $out/fuzz_seed${FUZZ_SEED:-616000}.log
VGL_CLI_FUZZ_CHUNKS=${CLI_CHUNKS:-20} VGL_CLI_FUZZ_SEED=${CLI_SEED:-616000} timeout -k 10 900 python -m pytest tests/test_gpu_cli_fuzz.py -m gpu -x -q > $out/clifuzz_seed${CLI_SEED:-616000}.log 2>&1 || { tail -40 $out/clifuzz_seed${CLI_SEED:-616000}.log; exit 1; }
tail -2 $out/clifuzz_seed${CLI_SEED:-616000}.log
