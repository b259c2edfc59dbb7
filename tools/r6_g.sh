#!/bin/bash
# round 6: dz stores between barriers A and B (k_dlstm_bwd), uploads on the upload stream (host-fed loop), RSRGAN_DPIPE default
cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests/test_gpu_placement.py tests/test_gpu_padrows.py tests/test_gpu_dist.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -8
run() { tag=$1; shift; env "$@" bash tools/prof.sh $tag --steps 6 --warmup 3 --no-variants --no-kernel-timing --no-cpu-baseline > /dev/null 2>&1
  echo "$tag: $(grep -E 'k_dlstm_bwd' gpurun_out/prof_$tag/r_kernel_stats.csv | cut -d, -f2-4 | tr '\n' ' ') | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_$tag/bench.log | head -1)"; }
run g0 RSRGAN_DW_INKERNEL=0
run g1 RSRGAN_DW_INKERNEL=1
for i in 1 2; do for g in 0 1; do
RSRGAN_DW_INKERNEL=$g timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/g_$g.log 2>&1
echo "dw_inkernel=$g: $(tail -1 gpurun_out/g_$g.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done; done
timeout 900 python bench.py --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/g_bench.json 2> gpurun_out/g_bench.err
python - <<'PY'
import json
b=json.loads(open('gpurun_out/g_bench.json').read().strip().splitlines()[-1])
print(b["value"], b["ms_per_step"], b["roofline"]["frac"])
for v in b.get("variants",[]):
    print(v["workload"][:95], v.get("ms_per_step"), {k:v[k] for k in v if k.startswith("RSRGAN") or k in ("error",)})
PY
