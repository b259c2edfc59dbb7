#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" bash tools/prof.sh $tag --steps 6 --warmup 3 --no-variants --no-kernel-timing --no-cpu-baseline > /dev/null 2>&1
  echo "$tag: $(grep -E 'k_dlstm_bwd' gpurun_out/prof_$tag/r_kernel_stats.csv | cut -d, -f2-4 | tr '\n' ' ') | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_$tag/bench.log | head -1)"; }
run d0 RSRGAN_DW_INKERNEL=0
run d1 RSRGAN_DW_INKERNEL=1
run d1dbg10 RSRGAN_DW_INKERNEL=1 RSRGAN_DW_DBG=10
timeout 600 python -m pytest tests/test_gpu_placement.py -k "weight_gradients_inside" -m gpu -x -q 2>&1 | tail -3
for g in 0 1; do
RSRGAN_DW_INKERNEL=$g timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/d_$g.log 2>&1
echo "dw_inkernel=$g: $(tail -1 gpurun_out/d_$g.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done
