#!/bin/bash
# the BASELINE-named network (2x512 unprojected + D=dnn) on the persistent path: kernel stats + timeline at B=64 and B=32
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for b in 64 32; do
bash tools/prof.sh x_bn$b --net baseline_named --batch $b --steps 5 --warmup 2 --no-variants --no-kernel-timing > /dev/null 2>&1
head -12 gpurun_out/prof_x_bn$b/r_kernel_stats.csv | cut -c1-130
f=$(find gpurun_out/prof_x_bn$b -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/timeline.py $f 3 > gpurun_out/timeline_x_bn$b.txt 2>&1
done
