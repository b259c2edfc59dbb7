#!/bin/bash
# the one-launch trailing form: same-box A/B, then the whole GPU suite
cd $GRAFT_REPO_ROOT
for i in 1 2; do for g in 0 1; do
RSRGAN_TRAIL=$g timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/t6_bench$g.log 2>&1
echo "trail=$g: $(tail -1 gpurun_out/t6_bench$g.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done; done
timeout 2900 python -m pytest tests -m gpu -q > gpurun_out/full_tests.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" gpurun_out/full_tests.log | grep -E "^E  |FAILED|passed|failed" | head -30
