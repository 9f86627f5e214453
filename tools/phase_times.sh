#!/bin/bash
# diagnostic: k_sample<2> launch time when the kernel returns after phase n (VGL_DEBUG_PHASE; the stamped DBG
# instantiation, so the differences between phases are the information, not the absolute times)
export VGL_LIB=$PWD/vcfgl_amd/lib/libvcfgl_hip_hooks.so   # VGL_DEBUG_PHASE exists in the -DVGL_TEST_HOOKS build only
for ph in 1 2 3 4 0; do
  echo -n "phase $ph: "
  VGL_DEBUG_PHASE=$ph timeout -k 10 200 python bench.py --no-cpu-baseline --sites 131072 --steps 2 2>/dev/null | grep -o 'avg_launch_ms": [0-9.]*\|"k_sample": [0-9.]*' | tr '\n' ' '; echo
done
