#!/bin/bash
# HBM traffic of one bench step from PMC counters, separate passes (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE in KB,
# FETCH_SIZE under-counts wide streaming reads by 2x on gfx950).  usage: tools/traffic.sh <tag> <bench args...>
tag=$1; shift
for c in FETCH_SIZE WRITE_SIZE; do
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$c
  mkdir -p $out
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -o r -- python $GRAFT_REPO_ROOT/bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-hbm-activity --no-kernel-timing > $out/bench.log 2>&1)
done
python - <<PY
import csv, glob, json, collections
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_%s/*counter_collection.csv" % c)
    tot = 0.0; per = collections.Counter(); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0][-44:]
        v = float(r["Counter_Value"]); tot += v; per[k] += v; cnt[k] += 1
    res[c] = {"total_KB_per_step": tot, "top": [(k, round(v), cnt[k]) for k, v in per.most_common(8)]}
json.dump(res, open("$GRAFT_REPO_ROOT/gpurun_out/traffic_${tag}.json", "w"), indent=1)
print(json.dumps(res)[:1500])
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_FETCH_SIZE $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_WRITE_SIZE
