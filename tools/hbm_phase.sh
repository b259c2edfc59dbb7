#!/bin/bash
# usage: tools/hbm_phase.sh <label> <command...> : runs the command (which should keep the GPU busy with ONE phase of the step for a few
# seconds) while rocm-smi's memory read/write activity (= amd-smi UMC_ACTIVITY, 1 % resolution) is sampled four times a second;
# prints the mean of the samples taken while it ran (first and last dropped) and x 8 TB/s
label=$1; shift
tmp=$(mktemp)
( while true; do rocm-smi --showmemuse 2>/dev/null | grep -o "Read/Write Activity (%): *[0-9.]*" | grep -o "[0-9.]*$" >> $tmp; sleep 0.2; done ) &
spid=$!
"$@" > ${tmp}.out 2>&1
kill $spid 2>/dev/null; wait $spid 2>/dev/null
python3 - "$label" $tmp ${tmp}.out <<'PY'
import sys
label, f, o = sys.argv[1:4]
v = [float(x) for x in open(f).read().split()]
mid = v[2:-1] if len(v) > 5 else v
m = sum(mid) / max(len(mid), 1)
tail = [l.strip() for l in open(o).read().splitlines() if l.strip()][-2:]
print("%-28s UMC activity %5.1f %% over %2d samples = %.2f TB/s | %s" % (label, m, len(mid), m / 100 * 8.0, " | ".join(t[:110] for t in tail)))
PY
rm -f $tmp ${tmp}.out
