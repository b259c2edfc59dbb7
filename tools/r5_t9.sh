#!/bin/bash
# D(G(x)) inside the generator's forward launch (G-runs that recompute the forward): same-box A/B on the 1D+2G schedules
cd $GRAFT_REPO_ROOT
for i in 1 2; do for g in 0 1; do
RSRGAN_TRAIL_FWD=$g timeout 300 python bench.py --gen-updates 2 --steps 30 --warmup 8 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/t9_a$g.log 2>&1
echo "lstm B=64 1D+2G fwd-trail=$g: $(tail -1 gpurun_out/t9_a$g.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
RSRGAN_TRAIL_FWD=$g timeout 300 python bench.py --net res_lstm_l --batch 8 --gen-updates 2 --steps 30 --warmup 8 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/t9_b$g.log 2>&1
echo "res_lstm_l B=8 1D+2G fwd-trail=$g: $(tail -1 gpurun_out/t9_b$g.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done; done
bash tools/prof.sh t9 --gen-updates 2 --steps 4 --warmup 2 --no-variants --no-kernel-timing > /dev/null 2>&1
grep -E "k_glstm|k_dlstm" gpurun_out/prof_t9/r_kernel_stats.csv | cut -c1-150
