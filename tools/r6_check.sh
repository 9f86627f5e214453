# usage (GPU box): bash tools/r6_check.sh <tag> <ab libs...>  -> the -m gpu suite, a 2-rank rehearsal of the N-rank bench on this one GPU (gloo), A/B of the named libraries
set -e
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/$tag
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest.log 2>&1 || { tail -40 gpurun_out/$tag/pytest.log; exit 1; }
tail -2 gpurun_out/$tag/pytest.log
timeout -k 10 600 python bench.py --gpus 2 --share-gpu --backend gloo --sites 131072 --steps 2 --warmup 1 --no-cpu-baseline --no-extra --records-leg-tiles 1 > gpurun_out/$tag/bench_2rank.json 2> gpurun_out/$tag/bench_2rank.err || { tail -20 gpurun_out/$tag/bench_2rank.err; exit 1; }
tail -1 gpurun_out/$tag/bench_2rank.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('value','value_with_record_gather','record_gather_leg','n_gpus')}); print(d['comm'])"
if [ $# -gt 0 ]; then AB_WORKLOADS="${AB_WORKLOADS:-c3 fixedq c4}" bash tools/ab_time.sh "$@" > gpurun_out/$tag/ab.txt 2>&1; cat gpurun_out/$tag/ab.txt; fi
