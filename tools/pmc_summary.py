#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs (one pass per counter group, as the MI355X
guide prescribes) into profiles/<tag>_pmc_summary.json and profiles/pmc_traffic.json.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are in KiB and on
gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md,
section HBM).  The raw (uncorrected) fetch figure is kept beside it: this path's loads are 1-8 bytes
per lane, a width the guide marks as uncalibrated.
profiles/pmc_traffic.json is keyed by bench workload: {workload: {"source": tag, "kernels": {kernel: {...}}}} -- bench.py
quotes it (labelled with its source) beside the numbers it measures itself.
usage: tools/pmc_summary.py <dir with FETCH_SIZE/ WRITE_SIZE/ SQ/ subdirs> <tag> [workload=c3] [sites per launch=65536]"""
import collections
import csv
import glob
import json
import os
import sys

root, tag = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "c3"
sites_per_launch = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
out = collections.defaultdict(dict)
# one file per counter group: the NEWEST (gpurun merges a call's files into gpurun_out/ beside those of an earlier call under the same tag -- round 6: a
# stale pass of the previous build sorted last and overrode the new one)
_newest = {}
for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    g = os.path.dirname(f)
    if g not in _newest or os.path.getmtime(f) > os.path.getmtime(_newest[g]):
        _newest[g] = f
for f in sorted(_newest.values(), key=lambda f: (os.sep + "SQ" + os.sep in f, f)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("k_"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            out[k]["vgpr"] = int(r["VGPR_Count"]); out[k]["sgpr"] = int(r["SGPR_Count"])
            out[k]["grid"] = int(r["Grid_Size"]); out[k]["lds"] = int(r["LDS_Block_Size"])
    for k, v in agg.items():
        for c, vals in v.items():
            out[k][c] = sum(vals) / len(vals)
            out[k]["launches_" + c] = len(vals)
def bucket(k):
    """the bench's kernel bucket of a kernel: k_sample_seg<DM, 1, LEAN> (round 6: one pool segment per wavefront) IS k_sample; its follow-up
    <DM, 2, LEAN> (the segment loop over listed wavefronts: nothing listed at the bench configurations) is kept under its own name"""
    if k.startswith("k_sample_seg<"):
        return "k_sample" if k.split(",")[1].strip() == "1" else "k_sample_seg_list"
    return k.split("<")[0]


traffic = {}
for k, v in out.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["hbm_bytes_per_launch"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
        v["hbm_bytes_per_launch_raw_fetch"] = (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
        traffic.setdefault(bucket(k), {}).update({"hbm_bytes_per_launch": v["hbm_bytes_per_launch"],
                                                        "raw_fetch_variant": v["hbm_bytes_per_launch_raw_fetch"]})
    if "SQ_ACTIVE_INST_VALU" in v and v.get("GRBM_GUI_ACTIVE"):
        # VALU pipes busy: SQ_ACTIVE_INST_VALU counts 4-cycle quads summed over the chip's 1024 SIMDs (256 CUs x 4);
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs (value / 8 / duration = 2.39 GHz, the part's clock), so one XCD's
        # busy cycles are value / 8 and each XCD holds 128 SIMDs.  Both counters come from the SAME pass (GRBM has its own two
        # slots beside the SQ's eight: tools/profile_round.sh), i.e. the same launches at the same clock -- round 2 divided by
        # the GRBM count of another pass, and short launches (k_depth, 0.7 ms) then read above 1.
        # SQ_ACTIVE_INST_VALU turns out to be one quad per VALU instruction (it tracks SQ_INSTS_VALU within 4 % on every kernel
        # here): the quotient is "VALU instructions x 4 cycles / SIMD cycles", an ISSUE-SLOT estimate.  The part issues its
        # two-operand 32-bit forms in about 2.6 cycles (tools/valu_rates.hip), so a kernel made of those (k_depth, the fixed-score
        # k_sample) reads above 1: the raw value is kept, the fraction is capped at 1.
        v["valu_issue_slots_at_4_cycles"] = 4.0 * v["SQ_ACTIVE_INST_VALU"] / (128.0 * v["GRBM_GUI_ACTIVE"])
        v["valu_busy_frac"] = min(1.0, v["valu_issue_slots_at_4_cycles"])
        traffic.setdefault(bucket(k), {}).update({"valu_busy_frac": v["valu_busy_frac"], "valu_issue_slots_at_4_cycles": v["valu_issue_slots_at_4_cycles"]})
    if "SQ_ACTIVE_INST_VALU" in v and v.get("SQ_WAVE_CYCLES"):
        # the kernel's own denominator: share of its wavefronts' resident cycles in which a VALU instruction of theirs was executing
        v["valu_active_share_of_wave_cycles"] = v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"]
    if v.get("SQ_INSTS_VALU") and v.get("SQ_WAVES"):            # (a kernel whose wavefronts leave at once -- k_sample_seg<., 2, .> with nothing listed -- has no vector instruction)
        v["valu_insts_per_wave"] = v["SQ_INSTS_VALU"] / v["SQ_WAVES"]
        v["active_lanes_per_valu_inst"] = v.get("SQ_THREAD_CYCLES_VALU", 0) / v["SQ_INSTS_VALU"]
        traffic.setdefault(bucket(k), {}).update({"valu_insts_per_wave": v["valu_insts_per_wave"],
                                                        "active_lanes_per_valu_inst": v["active_lanes_per_valu_inst"]})
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump(out, open(os.path.join(here, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
tpath = os.path.join(here, "profiles", "pmc_traffic.json")
try:
    allw = json.load(open(tpath))
except Exception:
    allw = {}
for e in traffic.values():
    e["sites_per_launch"] = sites_per_launch
sys.path.insert(0, here)
import bench
# SRC_SHA: the profiled tree's bench.source_sha() when the working tree has moved on since the run
allw[workload] = {"source": tag, "src_sha": os.environ.get("SRC_SHA") or bench.source_sha(), "kernels": traffic}
json.dump(allw, open(tpath, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
