#!/bin/bash
# tagged ring slots (no re-arming stores) against the sentinel form: harness, parity tests, same-box bench A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for g in 0 1 0 1; do
for m in "" b "b x"; do for n in 64 32; do GP_TAGS=$g timeout 60 tools/ubench/gpersist_trace_nt $n 100 3 $m | sed "s/^/tags=$g: /" | cut -c1-170; done; done
done
timeout 900 python -m pytest tests/test_gpu_placement.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -5
for i in 1 2; do for g in 0 1; do
RSRGAN_GP_TAGS=$g timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > gpurun_out/z_bench$g.log 2>&1
echo "tags=$g: $(tail -1 gpurun_out/z_bench$g.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
done; done
