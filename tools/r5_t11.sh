#!/bin/bash
# RSRGAN_DPIPE: how many CUs the tail's GEMMs leave to the next D(real)
cd $GRAFT_REPO_ROOT
echo "dpipe=0: $(RSRGAN_DPIPE=0 timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
for w in 256 224 208 192; do
echo "dpipe=1 w=$w: $(RSRGAN_DPIPE=1 RSRGAN_DPIPE_W=$w timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done
echo "dpipe=0: $(RSRGAN_DPIPE=0 timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
export RSRGAN_DPIPE=1 RSRGAN_DPIPE_W=224
bash tools/prof.sh t11 --steps 5 --warmup 2 --no-variants --no-kernel-timing > /dev/null 2>&1
f=$(find gpurun_out/prof_t11 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/timeline.py $f 2 > gpurun_out/timeline_t11.txt 2>&1
awk '$2>60' gpurun_out/timeline_t11.txt | cut -c1-100; tail -1 gpurun_out/timeline_t11.txt
