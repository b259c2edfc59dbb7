#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-a}
(cd tools/ubench && timeout 120 ./l2 > ../../gpurun_out/l2_$tag.txt 2>&1; sed -n '/== C/,$p' ../../gpurun_out/l2_$tag.txt)
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_l2$tag; mkdir -p $out
(cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $out -o r -- $GRAFT_REPO_ROOT/tools/ubench/l2 E > $out/run.log 2>&1)
tail -12 $out/run.log
python - <<PY
import csv, glob
f = glob.glob("$out/*counter_collection.csv")
if f:
    rows = list(csv.DictReader(open(f[0])))
    # per dispatch: kernel name, hit, miss ; print the last 120 dispatches compactly
    d = {}
    for r in rows:
        k = int(r["Dispatch_Id"]); d.setdefault(k, {"n": r["Kernel_Name"][:28]})[r["Counter_Name"]] = float(r["Counter_Value"])
    ks = sorted(d)
    with open("$out/per_dispatch.txt", "w") as o:
        for k in ks:
            o.write("%d %s hit=%.0f miss=%.0f\n" % (k, d[k]["n"], d[k].get("TCC_HIT_sum", -1), d[k].get("TCC_MISS_sum", -1)))
    print(len(ks), "dispatches")
else:
    print("no counter csv", glob.glob("$out/*"))
PY
rm -f $out/*counter_collection.csv $out/*kernel_trace.csv
timeout 600 python -m pytest tests/test_gpu_trainers.py -q -m gpu -k rced > gpurun_out/t_rced_$tag.log 2>&1; echo "pytest rced rc=$?"; tail -5 gpurun_out/t_rced_$tag.log
PYTHONFAULTHANDLER=1 timeout -s ABRT 500 python bench.py > gpurun_out/bench_default_$tag.log 2>&1; echo "bench rc=$?"; tail -3 gpurun_out/bench_default_$tag.log | cut -c1-3000
