# full-size oracle parity at the final build: bash tools/r6_full_parity.sh <case> [sites]
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r06_parity
timeout -k 10 1100 python tools/full_parity.py "$@" > gpurun_out/r06_parity/$1.log 2>&1 || { tail -5 gpurun_out/r06_parity/$1.log; exit 1; }
tail -2 gpurun_out/r06_parity/$1.log
