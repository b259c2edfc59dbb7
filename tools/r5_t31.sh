#!/bin/bash
# who paces k_glstm_bwd_dt of a padded batch (one lane, one tile): RSRGAN_TRAIL_DBG 2 = the discriminator half alone, 3 = both halves without the coupling
cd $GRAFT_REPO_ROOT
for m in 0 2 3; do
RSRGAN_TRAIL_DBG=$m bash tools/prof.sh t31_$m --net lstm --batch 8 --steps 6 --warmup 2 --no-variants --no-kernel-timing > /dev/null 2>&1
echo "TRAIL_DBG=$m: $(grep -E 'k_glstm_bwd_dt|k_glstm_fwd_dt' gpurun_out/prof_t31_$m/r_kernel_stats.csv | cut -d, -f1-4 | cut -c1-40,95-160 | tr '\n' ' ')"
done
