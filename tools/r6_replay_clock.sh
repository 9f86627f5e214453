# usage (GPU box): bash tools/r6_replay_clock.sh <replay name> <out dir>  -> the replay's effective shader clock from GRBM_GUI_ACTIVE (MI355X_MICROARCH.md, DVFS give-back)
export TMPDIR=/tmp
k=$1; out=$2; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/$k tools/replay/$k.hip 2>/dev/null || exit 1
cd /tmp && rm -rf /tmp/rp_$k && rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/rp_$k -- /tmp/$k > $OLDPWD/$out/${k}_pmc_run.json 2> /tmp/rp_$k.err; cd $OLDPWD
python3 - <<PY
import csv, glob
cc = glob.glob("/tmp/rp_$k/*/*counter_collection.csv")[0]
kt = glob.glob("/tmp/rp_$k/*/*kernel_trace.csv")[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r.get("Dispatch_Id")] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        d = dur.get(r.get("Dispatch_Id"))
        if d: print(f"dispatch {r['Dispatch_Id']} grid {r['Grid_Size']}: {d*1e3:.3f} ms, GRBM_GUI_ACTIVE {float(r['Counter_Value']):.0f} -> effective clock {float(r['Counter_Value'])/8/d/1e9:.3f} GHz")
PY
cat $out/${k}_pmc_run.json
