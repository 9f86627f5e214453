# usage (GPU box): bash tools/r6_replay_clock.sh <replay name> <out dir>
# Runs tools/replay/<name>.hip (tools/isa_hist.py --emit-replay) under rocprofv3 with GRBM_GUI_ACTIVE and writes <out dir>/<name>_replay.json:
# SIMD cycles per wavefront-iteration of the replayed loop = GRBM_GUI_ACTIVE / 8 of the eight-resident-set dispatch / (8 sets x wavefronts per SIMD x iterations)
# -- counted cycles, no clock assumed (MI355X_MICROARCH.md: GRBM_GUI_ACTIVE is summed over the 8 XCDs).
export TMPDIR=/tmp
k=$1; out=$2; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/$k tools/replay/$k.hip 2>/dev/null || { echo "compile of $k failed"; exit 1; }
here=$PWD
cd /tmp && rm -rf /tmp/rp_$k && rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/rp_$k -- /tmp/$k > $here/$out/${k}_run.json 2> /tmp/rp_$k.err; cd $here
python3 - <<PY
import csv, glob, json
run = json.loads(open("$out/${k}_run.json").read().strip().splitlines()[-1])
cc = glob.glob("/tmp/rp_$k/*/*counter_collection.csv")[0]
kt = glob.glob("/tmp/rp_$k/*/*kernel_trace.csv")[0]
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9 for r in csv.DictReader(open(kt))}
rows = [r for r in csv.DictReader(open(cc)) if r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
big = rows[-1]                                                   # the second of the two eight-set dispatches
grbm, d = float(big["Counter_Value"]), dur[big["Dispatch_Id"]]
cyc = grbm / 8.0 / (8 * run["waves_per_simd"] * run["iterations"])
res = dict(run, grbm_gui_active=grbm, dispatch_ms=d * 1e3, effective_clock_ghz=grbm / 8.0 / d / 1e9, simd_cycles_per_iteration_grbm=cyc,
           cycles_per_valu_inst_grbm=cyc / run["valu_per_iteration"],
           method="tools/isa_hist.py --emit-replay: the loop's vector + scalar instructions without memory / LDS / waits / branches, every SIMD at the kernel's occupancy, "
                  "eight resident sets back to back; cycles = GRBM_GUI_ACTIVE / 8 of that dispatch / wavefront-iterations per SIMD")
json.dump(res, open("$out/${k}_replay.json", "w"), indent=1)
print(json.dumps({k2: res[k2] for k2 in ("valu_per_iteration", "salu_per_iteration", "waves_per_simd", "simd_cycles_per_iteration_grbm", "cycles_per_valu_inst_grbm", "effective_clock_ghz", "simd_cycles_per_iteration_slowest_wave")}))
PY
