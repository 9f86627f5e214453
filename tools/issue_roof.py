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
rare = sum(b["weight"] * b["cycles"] for b in hist["blocks"] if 0 < b["weight"] < 0.5)        # the bounded tests: priced by the class rates, scaled like the main path
main_valu = rep["valu_per_iteration"]
scale = rep["simd_cycles_per_iteration_grbm"] / max(1e-9, sum(b["cycles"] for b in hist["blocks"] if b["weight"] >= 0.5))
entry = {"kernel": bucket, "kernel_build": hist["kernel"], "src_sha": bench.source_sha(),
         "issue_cycles_per_inst": rep["cycles_per_valu_inst_grbm"],
         "loop": {"valu_per_iteration_main_path": main_valu, "simd_cycles_per_iteration_main_path": rep["simd_cycles_per_iteration_grbm"],
                  "rare_blocks_cycles_per_iteration": rare * scale, "class_sum_cycles_main_path": sum(b["cycles"] for b in hist["blocks"] if b["weight"] >= 0.5),
                  "iterations_per_wave": st.get("pool iterations"), "lanes_with_item": st.get("per_iteration", {}).get("lanes_with_item")},
         "replay": {k: rep[k] for k in ("waves_per_simd", "iterations", "effective_clock_ghz", "simd_cycles_per_iteration_slowest_wave", "grbm_gui_active", "dispatch_ms")},
         "method": "cycles per vector instruction of the kernel's pool loop (main path), measured by replaying its instruction sequence on the box (tools/isa_hist.py --emit-replay, "
                   "GRBM_GUI_ACTIVE cycles); bench.py multiplies by the kernel's vector instructions per wavefront (committed counters)",
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
