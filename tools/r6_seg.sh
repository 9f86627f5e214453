# usage (GPU box): bash tools/r6_seg.sh <lib ...> -> parity cases of the split build, then same-box A/B of the named libraries (vcfgl_amd/lib_ab/<name>.so)
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r06_seg
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale_oracle.py -x -q -k "split_build or deferred or maximum_depth or undecided or c3 or c4 or qsi16 or alltags or adjbins or gl1q" > gpurun_out/r06_seg/pytest.log 2>&1 || { tail -30 gpurun_out/r06_seg/pytest.log; exit 1; }
tail -2 gpurun_out/r06_seg/pytest.log
AB_WORKLOADS="${AB_WORKLOADS:-c3 c4 qsi16 alltags}" bash tools/ab_time.sh "$@" > gpurun_out/r06_seg/ab.txt 2>&1
cat gpurun_out/r06_seg/ab.txt
