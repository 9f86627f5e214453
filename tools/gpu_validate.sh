# usage (GPU box): bash tools/gpu_validate.sh  -> the whole -m gpu suite, a long fuzz run with fresh seeds, the host-program fuzz, full-size parity one-offs
set -e
export TMPDIR=/tmp
out=gpurun_out/r4val; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
VGL_FUZZ_CHUNKS=${FUZZ_CHUNKS:-200} VGL_FUZZ_SEED=${FUZZ_SEED:-404000} timeout -k 10 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k random_configurations > $out/fuzz.log 2>&1 || { tail -40 $out/fuzz.log; exit 1; }
tail -2 $out/fuzz.log
VGL_CLI_FUZZ_CHUNKS=${CLI_CHUNKS:-30} VGL_CLI_FUZZ_SEED=${CLI_SEED:-404000} timeout -k 10 1200 python -m pytest tests/test_gpu_cli_fuzz.py -m gpu -x -q > $out/clifuzz.log 2>&1 || { tail -40 $out/clifuzz.log; exit 1; }
tail -2 $out/clifuzz.log
