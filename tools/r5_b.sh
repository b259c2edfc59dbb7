#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for m in "X=1" "HSA_CU_MASK=0:0-127" "ROC_GLOBAL_CU_MASK=0xffffffffffffffffffffffffffffffff"; do
  env $m python -c "import torch; print('$m', torch.cuda.get_device_properties(0).multi_processor_count)" 2>&1 | tail -1
done > gpurun_out/b_masks.log 2>&1
cat gpurun_out/b_masks.log
( for v in p0 p1 p2 p3 p3g0 p3g14 p3g34 p3g11 p1g14; do
    for mode in "" b; do
      echo -n "$v $mode: "; timeout 60 tools/ubench/gpersist_trace_nt_$v 64 100 3 $mode | head -1
    done
  done
  for v in p0 p3; do for mode in "" b; do echo -n "$v N=32 $mode: "; timeout 60 tools/ubench/gpersist_trace_nt_$v 32 50 3 $mode | head -1; done; done
) > gpurun_out/b_harness.log 2>&1
cat gpurun_out/b_harness.log | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_placement.py -x -q -m gpu -p no:cacheprovider -k "progressive" > gpurun_out/b_tests.log 2>&1; tail -5 gpurun_out/b_tests.log
for p in 0 1 2 3; do
  RSRGAN_GP_PROG=$p timeout 300 python bench.py --steps 30 --warmup 8 --no-variants --no-cpu-baseline --no-hbm-activity > gpurun_out/b_bench_p$p.log 2>&1; echo "PROG=$p $(tail -1 gpurun_out/b_bench_p$p.log | cut -c1-200)"
done
