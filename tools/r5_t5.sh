#!/bin/bash
cd $GRAFT_REPO_ROOT
for d in 1 3 2; do
export RSRGAN_TRAIL_DBG=$d
bash tools/prof.sh t5 --steps 5 --warmup 2 --no-variants --no-kernel-timing > /dev/null 2>&1
echo "dbg=$d $(grep -E "k_glstm_bwd_dt" gpurun_out/prof_t5/r_kernel_stats.csv | cut -c1-140)"
done
