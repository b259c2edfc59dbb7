#!/bin/bash
# RSRGAN_DPIPE: parity (async sequence), same-box A/B of the headline, timeline
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_placement.py -k "pipelined_discriminator" -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -15
for i in 1 2; do for g in 0 1; do
RSRGAN_DPIPE=$g timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/t10_$g.log 2>&1
echo "dpipe=$g: $(tail -1 gpurun_out/t10_$g.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done; done
export RSRGAN_DPIPE=1
bash tools/prof.sh t10 --steps 5 --warmup 2 --no-variants --no-kernel-timing > /dev/null 2>&1
f=$(find gpurun_out/prof_t10 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/timeline.py $f 2 > gpurun_out/timeline_t10.txt 2>&1
awk '$2>15' gpurun_out/timeline_t10.txt | cut -c1-100; tail -1 gpurun_out/timeline_t10.txt
