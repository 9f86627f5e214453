# usage (GPU box): bash tools/r6_replay.sh <out dir> <replay name ...>   -> compiles and runs tools/replay/<name>.hip with the box's current sclk
out=$1; shift; mkdir -p $out
for k in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/$k tools/replay/$k.hip 2>/dev/null || { echo "compile of $k failed"; exit 1; }
  ( while true; do rocm-smi --showclocks 2>/dev/null | grep -E "sclk" | head -1; sleep 0.2; done ) > $out/$k.clk 2>&1 &
  W=$!
  /tmp/$k > $out/$k.json; /tmp/$k >> $out/$k.json
  kill $W 2>/dev/null
  cat $out/$k.json
done
