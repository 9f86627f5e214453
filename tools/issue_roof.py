#!/usr/bin/env python3
"""profiles/issue_roof.json: per bench workload, the SIMD cycles per vector instruction of the dominant kernel's hot loop (what bench.py's roofline.issue_* is priced
with).  Inputs, all committed under profiles/: the loop's histogram (tools/isa_hist.py --json), its replay on the GPU box (tools/r6_replay_clock.sh), the stamped
build's per-iteration counts (tools/stamps.py 2).
usage: tools/issue_roof.py <workload[,workload...]> <isa_hist json> <replay json> <stamps json> [kernel bucket=k_sample]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
wls, hist, rep, st = sys.argv[1].split(","), json.load(open(sys.argv[2])), json.load(open(sys.argv[3])), json.load(open(sys.argv[4]))
bucket = sys.argv[5] if len(sys.argv) > 5 else "k_sample"
rare_class = sum(b["weight"] * b["cycles"] for b in hist["blocks"] if 0 < b["weight"] < 0.5)  # the bounded tests at the class rates (kept for the record: the classes do not add, so this is NOT a lower bound)
rare_valu = sum(b["weight"] * b["valu"] for b in hist["blocks"] if 0 < b["weight"] < 0.5)
# the cheapest a vector instruction can be on this part (v_fma_f32 / v_mov_b32 / v_add_u32 at eight wavefronts per SIMD: tools/valu_rates.hip): what the
# kernel's instructions OUTSIDE the replayed loop are priced at, so that the sum stays a lower bound of the time
floor = min(float(l.split()[3]) for l in open(os.path.join(ROOT, "profiles", "r06_valu_rates.txt")) if l.startswith(("k_fma_f32 ", "k_mov ", "k_add_u32 ")) and "waves/SIMD 8" in l)
main_valu = rep["valu_per_iteration"]
scale = rep["simd_cycles_per_iteration_grbm"] / max(1e-9, sum(b["cycles"] for b in hist["blocks"] if b["weight"] >= 0.5))
# second session of round 6: the rarely-run blocks (the two bounded tests) are priced at the FLOOR rate too, like everything outside the replayed path -- every
# term of a lower bound has to be one, and their class prices scaled like the main path were not (with the kernel 2 % faster the sum read 1.01)
rare = rare_valu * floor
entry = {"kernel": bucket, "kernel_build": hist["kernel"], "src_sha": bench.source_sha(),
         "issue_cycles_per_inst": rep["cycles_per_valu_inst_grbm"],
         "loop": {"valu_per_iteration_main_path": main_valu, "simd_cycles_per_iteration_main_path": rep["simd_cycles_per_iteration_grbm"],
                  "rare_blocks_cycles_per_iteration": rare, "rare_blocks_cycles_per_iteration_at_class_rates": rare_class * scale, "rare_blocks_valu_per_iteration": rare_valu, "floor_cycles_per_inst_outside_the_loop": floor,
                  "class_sum_cycles_main_path": sum(b["cycles"] for b in hist["blocks"] if b["weight"] >= 0.5),
                  "iterations_per_wave": st.get("pool iterations"), "lanes_with_item": st.get("per_iteration", {}).get("lanes_with_item")},
         "replay": {k: rep[k] for k in ("waves_per_simd", "iterations", "effective_clock_ghz", "simd_cycles_per_iteration_slowest_wave", "grbm_gui_active", "dispatch_ms")},
         "method": "SIMD cycles the kernel's vector instructions need, per wavefront = pool iterations (stamped build) x [the loop's every-iteration path, replayed on the box "
                   "with its own instruction sequence (tools/isa_hist.py --emit-replay, GRBM_GUI_ACTIVE cycles, the fastest of five runs)] + the instructions of its "
                   "rarely-run blocks and outside the loop (counters' total minus the replayed path's) x the cheapest issue rate of the part: a LOWER bound of the kernel's time",
         "source": "profiles/" + os.path.basename(sys.argv[3])}
path = os.path.join(ROOT, "profiles", "issue_roof.json")
try:
    allw = json.load(open(path))
except Exception:
    allw = {}
for w in wls:
    allw[w] = entry
json.dump(allw, open(path, "w"), indent=1, sort_keys=True)
print(json.dumps(entry, indent=1))
