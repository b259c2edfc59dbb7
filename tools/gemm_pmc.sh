#!/bin/bash
# usage: tools/gemm_pmc.sh <tag> "<counters>"  : one rocprofv3 --pmc pass over `tools/ubench/gemm_bench abl` (k_gemm next to hipBLASLt on
# 4096^3 and 560 x 3040 x 6400); per-kernel means -> gpurun_out/gemm_pmc_<tag>.txt
tag=$1; ctrs=$2
out=$GRAFT_REPO_ROOT/gpurun_out/gemm_pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o r -- $GRAFT_REPO_ROOT/tools/ubench/gemm_bench abl > $out/run.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, collections, glob
f = glob.glob("$out/*counter_collection.csv")
if not f:
    print("no counter csv", glob.glob("$out/*")); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0][:70]
    if "k_ref" in k or "fill" in k: continue
    key = (k, r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(key, r["Counter_Name"])] += 1
with open("$GRAFT_REPO_ROOT/gpurun_out/gemm_pmc_$tag.txt", "w") as o:
    for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
        line = "%s grid=%s wg=%s | " % k + " ".join("%s=%.4g(n=%d)" % (c, v / cnt[(k, c)], cnt[(k, c)]) for c, v in sorted(agg[k].items()))
        print(line); o.write(line + "\n")
PY
rm -rf $out
