#!/bin/bash
# trailing discriminator BPTT: parity + same-box A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_placement.py -k "trailing" -m gpu -x -q 2>&1 | tail -15
for i in 1 2; do for g in 0 1; do
RSRGAN_TRAIL=$g timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/t1_bench$g.log 2>&1
echo "trail=$g: $(tail -1 gpurun_out/t1_bench$g.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done; done
tail -3 gpurun_out/t1_bench1.log | cut -c1-600
