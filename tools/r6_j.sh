#!/bin/bash
# round 6: the input FC's parameter gradients between the projection gradients and the column sums on the side stream (RSRGAN_POST_MID)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "test_full_size_step_against_oracle or res_lstm_l-32" 2>&1 | tail -2
for i in 1 2 3; do for g in 0 1; do
RSRGAN_POST_MID=$g timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/j_$g.log 2>&1
echo "post_mid=$g: $(tail -1 gpurun_out/j_$g.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done; done
bash tools/prof.sh r6j --steps 5 --warmup 2 --no-variants --no-kernel-timing > /dev/null 2>&1
f=$(find gpurun_out/prof_r6j -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/timeline.py $f 2 > gpurun_out/timeline_r6j.txt 2>&1
awk '$1>3800' gpurun_out/timeline_r6j.txt | cut -c1-100
