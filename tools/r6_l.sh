#!/bin/bash
# round 6 experiment: plain (L2-scope) granule stores in k_dlstm_bwd (a tile's workgroups share an XCD at 128 rows) vs write-through
cd $GRAFT_REPO_ROOT
cp rsrgan_amd/lib/librsrgan_hip.so /tmp/keep.so
for L in base fast base fast; do
  cp tools/ab_libs/lib_$L.so rsrgan_amd/lib/librsrgan_hip.so
  bash tools/prof.sh l$L --steps 6 --warmup 3 --no-variants --no-kernel-timing --no-cpu-baseline > /dev/null 2>&1
  echo "$L: $(grep -E 'k_dlstm_bwd' gpurun_out/prof_l$L/r_kernel_stats.csv | cut -d, -f2-4) | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_l$L/bench.log | head -1) | $(grep -o 'losses_last_step": \[[^]]*' gpurun_out/prof_l$L/bench.log | head -1)"
done
cp tools/ab_libs/lib_fast.so rsrgan_amd/lib/librsrgan_hip.so
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "test_full_size_step_against_oracle" 2>&1 | tail -3
cp /tmp/keep.so rsrgan_amd/lib/librsrgan_hip.so
