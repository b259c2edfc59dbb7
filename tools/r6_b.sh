#!/bin/bash
# round 6: the discriminator's weight gradients inside k_dlstm_bwd (RSRGAN_DW_INKERNEL): parity vs the launch form, same-box A/B, timeline
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_placement.py -k "weight_gradients_inside" -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -15
for i in 1 2; do for g in 0 1; do
RSRGAN_DW_INKERNEL=$g timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/b_$g.log 2>&1
echo "dw_inkernel=$g: $(tail -1 gpurun_out/b_$g.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done; done
bash tools/prof.sh r6b --steps 5 --warmup 2 --no-variants --no-kernel-timing > /dev/null 2>&1
f=$(find gpurun_out/prof_r6b -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/timeline.py $f 2 > gpurun_out/timeline_r6b.txt 2>&1
awk '$2>3' gpurun_out/timeline_r6b.txt | cut -c1-100 | head -70; tail -1 gpurun_out/timeline_r6b.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_padrows.py tests/test_gpu_parity.py tests/test_gpu_dist.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -8
