#!/bin/bash
# HBM-side bytes of one step with WARM L2s: the plain PMC passes (tools/traffic.sh) instrument -- and thereby serialise -- every
# dispatch, so each kernel starts with cold L2s and the sum is an upper bound.  Here each pass collects FETCH_SIZE / WRITE_SIZE for ONE
# kernel class only (--kernel-include-regex); the other launches run un-instrumented in between, as in the un-profiled step, so the
# instrumented launches see the L2 contents their predecessors left.  usage: tools/traffic_warm.sh <tag> <bench args...>
tag=$1; shift
classes="k_fwd_gates k_fwd_proj k_bwd_a2 k_bwd_bp k_bwd_b_red k_bwd_b< k_gemm k_colsum|k_lstm_colsums k_apply|k_swizzle|k_transpose"
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/tw_$tag
for c in FETCH_SIZE WRITE_SIZE; do
  i=0
  for cls in $classes; do
    i=$((i+1))
    out=$GRAFT_REPO_ROOT/gpurun_out/tw_$tag/${c}_$i
    mkdir -p $out
    (cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --pmc $c --kernel-include-regex "$cls" --kernel-trace --output-format csv -d $out -o r -- python $GRAFT_REPO_ROOT/bench.py "$@" --steps 3 --warmup 2 --no-cpu-baseline --no-hbm-activity --no-kernel-timing --no-variants > $out/bench.log 2>&1)
  done
done
python - <<PY
import csv, glob, json, collections
root = "$GRAFT_REPO_ROOT/gpurun_out/tw_$tag"
res = {"method": "one rocprofv3 --pmc pass per (counter, kernel class) with --kernel-include-regex: only that class is instrumented, the "
                 "rest of the step runs as un-profiled; KB per step = per-class sum over the run / steps in the run (5: 2 warm-up + 3)",
       "classes": {}}
steps = 5.0
tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for d in sorted(glob.glob(root + "/%s_*" % c)):
        f = glob.glob(d + "/*counter_collection.csv")
        if not f:
            continue
        per = collections.Counter(); cnt = collections.Counter()
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] != c: continue
            k = r["Kernel_Name"].split("(")[0][-40:]
            per[k] += float(r["Counter_Value"]); cnt[k] += 1
        for k, v in per.items():
            e = res["classes"].setdefault(k, {})
            e[c + "_KB_per_step"] = round(v / steps, 1); e["launches_per_step"] = round(cnt[k] / steps, 1)
            tot[c] += v / steps
res["FETCH_SIZE_KB_per_step"] = round(tot["FETCH_SIZE"], 1); res["WRITE_SIZE_KB_per_step"] = round(tot["WRITE_SIZE"], 1)
res["hbm_bytes_per_step_warm"] = int((2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024)
json.dump(res, open("$GRAFT_REPO_ROOT/gpurun_out/traffic_warm_$tag.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "classes"}))
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/tw_$tag
