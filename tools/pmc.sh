#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" <bench args...>  : one rocprofv3 --pmc pass (no other tracing domains)
tag=$1; shift; ctrs=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o r -- python $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline --no-hbm-activity > $out/bench.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, collections, glob
f = glob.glob("$out/*counter_collection.csv")
if not f:
    print("no counter csv", glob.glob("$out/*")); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0][-40:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    cnt[(k, r["Counter_Name"])] += 1
with open("$out/summary.txt", "w") as o:
    for k in sorted(agg, key=lambda k: -sum(agg[k].values()))[:14]:
        line = k + " | " + " ".join("%s=%.4g(n=%d)" % (c, v / cnt[(k, c)], cnt[(k, c)]) for c, v in sorted(agg[k].items()))
        print(line); o.write(line + "\n")
PY
rm -f $out/*counter_collection.csv $out/*kernel_trace.csv
