# usage (GPU box): bash tools/clock_watch.sh <out file> -- <command ...>   samples rocm-smi clocks / power twice a second while the command runs
out=$1; shift; shift
( while true; do rocm-smi --showclocks --showpower --showperflevel 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | tr '\n' ' ' ; echo; sleep 0.5; done ) > "$out" 2>&1 &
W=$!
"$@"
rc=$?
kill $W 2>/dev/null
exit $rc
