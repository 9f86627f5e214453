#!/bin/bash
# usage (development container): tools/r6_issue_roof_cpu.sh  -> the pool loop's histograms and replay programs of the CURRENT build (C3, depth 30 with bins,
# the optional-tag build) from the compiler's assembly: profiles/r06_isa_hist_*.json, tools/replay/replay_*.hip.  Then, on the GPU box, tools/r6_replay_clock.sh
# per replay, and tools/issue_roof.py per workload (tools/r6_issue_roof_collect.sh).
set -eu
make -s -C vcfgl_amd/csrc asm > /dev/null 2>&1
A=build/asm/vgl_sample_seg-hip-amdgcn-amd-amdhsa-gfx950.s
R=profiles/r06_valu_rates.txt
python tools/isa_hist.py --asm $A --kernel 'k_sample_seg<1, 1, 2>' --rates $R --auto profiles/r06_ab/stamps_lean2_depth20.json --bins 0 --json profiles/r06_isa_hist_c3.json --emit-replay tools/replay/replay_c3.hip | tail -3
python tools/isa_hist.py --asm $A --kernel 'k_sample_seg<1, 1, 2>' --rates $R --auto profiles/r06_ab/stamps_lean2_depth30.json --bins 1 --json profiles/r06_isa_hist_c4.json --emit-replay tools/replay/replay_c4.hip | tail -3
python tools/isa_hist.py --asm $A --kernel 'k_sample_seg<1, 1, 4>' --rates $R --auto profiles/r06_ab/stamps_lean2_depth20.json --bins 0 --json profiles/r06_isa_hist_qsi16.json --emit-replay tools/replay/replay_qsi16.hip | tail -3
