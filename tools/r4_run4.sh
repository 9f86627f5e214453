set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r4d
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${K:-optional or deferred}" > gpurun_out/r4d/pytest.log 2>&1 || { tail -40 gpurun_out/r4d/pytest.log; exit 1; }
tail -3 gpurun_out/r4d/pytest.log
