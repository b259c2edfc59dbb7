#!/bin/bash
# trailing discriminator BPTT, one launch / two launches / off: parity, timeline, same-box A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_placement.py -k "trailing" -m gpu -x -q 2>&1 | tail -5
bash tools/prof.sh t2 --steps 5 --warmup 2 --no-variants --no-kernel-timing > /dev/null 2>&1
f=$(find gpurun_out/prof_t2 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/timeline.py $f 2 > gpurun_out/timeline_t2.txt 2>&1
sed -n 30,40p gpurun_out/timeline_t2.txt; tail -1 gpurun_out/timeline_t2.txt
for i in 1 2; do for g in 0 1; do
RSRGAN_TRAIL=$g timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/t1_bench$g.log 2>&1
echo "trail=$g: $(tail -1 gpurun_out/t1_bench$g.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done; done
