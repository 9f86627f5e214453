# round 5, first GPU call: the -m gpu suite, the driver's own bench command (is the last stdout line small and parseable?), and the
# RCCL group code at world 1 (BENCH_FORCE_DIST) -- bash tools/r5_first.sh <tag>
set -e
export TMPDIR=/tmp
tag=${1:-r05_a}
mkdir -p gpurun_out/$tag
timeout -k 10 900 python -m pytest ${PYT:-tests} -m gpu -x -q > gpurun_out/$tag/pytest.log 2>&1 || { tail -40 gpurun_out/$tag/pytest.log; exit 1; }
tail -3 gpurun_out/$tag/pytest.log
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail-file gpurun_out/$tag/bench_detail.json > gpurun_out/$tag/bench.out 2> gpurun_out/$tag/bench.err
python3 - <<PY
import json
ls=open("gpurun_out/$tag/bench.out").read().strip().splitlines()
print("stdout lines", len(ls), [len(l) for l in ls])
d=json.loads(ls[-1]); print("value %.4e"%d["value"], "frac", d["roofline"]["frac"], "cpu", d.get("cpu_baseline",{}).get("value"), d["extra"])
PY
BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --gpus 1 --steps 3 --warmup 1 --no-extra --no-cpu-baseline --sites 131072 --detail-file gpurun_out/$tag/dist1_detail.json > gpurun_out/$tag/dist1.out 2> gpurun_out/$tag/dist1.err
tail -c 1500 gpurun_out/$tag/dist1.out
