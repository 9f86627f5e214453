# usage (GPU box): bash tools/r6_final_roof.sh [repetitions]  -> the three replay programs of the current build under GRBM_GUI_ACTIVE, several times
# (gpurun_out/r06_v3_replay/rep<i>/*_replay.json): the roof takes the FASTEST replay of each loop -- it is a lower bound
set -e
for i in $(seq 1 ${1:-5}); do for k in replay_c3 replay_c4 replay_qsi16; do bash tools/r6_replay_clock.sh $k gpurun_out/r06_v3_replay/rep$i | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$k rep$i', d['simd_cycles_per_iteration_grbm'], d['effective_clock_ghz'])"; done; done
