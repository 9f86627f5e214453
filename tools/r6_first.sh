set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r06_base
ls /sys/class/drm/ > gpurun_out/r06_base/sysfs.txt 2>&1 || true
for c in /sys/class/drm/card*/device; do echo $c; cat $c/pp_dpm_sclk $c/pp_dpm_mclk 2>&1 | head -20; done >> gpurun_out/r06_base/sysfs.txt 2>&1 || true
python -c "import amdsmi; print('amdsmi ok')" >> gpurun_out/r06_base/sysfs.txt 2>&1 || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rates tools/valu_rates.hip 2>/dev/null
/tmp/valu_rates > gpurun_out/r06_base/valu_rates.txt
python bench.py > gpurun_out/r06_base/bench.json 2> gpurun_out/r06_base/bench.err
tail -c 2500 gpurun_out/r06_base/bench.json
