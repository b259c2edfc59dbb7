#!/bin/bash
# round 6: what the chain side of the in-launch weight gradients costs: k_dlstm_bwd per build variant (tools/ab_libs/lib_exp<e>.so, -DDW_EXP=e:
# 0 product | 1 plain dz stores | 2 plain stores, no progress word | 3 no dz stores | 4 no dz stores, no word; weight-gradient workgroups idle in 1..4)
cd $GRAFT_REPO_ROOT
cp rsrgan_amd/lib/librsrgan_hip.so /tmp/keep.so
for e in 0 1 2 3 4 0; do
  cp tools/ab_libs/lib_exp$e.so rsrgan_amd/lib/librsrgan_hip.so
  bash tools/prof.sh h$e --steps 6 --warmup 3 --no-variants --no-kernel-timing --no-cpu-baseline > /dev/null 2>&1
  echo "exp$e: $(grep -E 'k_dlstm_bwd' gpurun_out/prof_h$e/r_kernel_stats.csv | cut -d, -f2-4 | tr '\n' ' ')"
done
cp /tmp/keep.so rsrgan_amd/lib/librsrgan_hip.so
RSRGAN_DW_INKERNEL=0 bash tools/prof.sh hb --steps 6 --warmup 3 --no-variants --no-kernel-timing --no-cpu-baseline > /dev/null 2>&1
echo "baseline (DW_INKERNEL=0): $(grep -E 'k_dlstm_bwd' gpurun_out/prof_hb/r_kernel_stats.csv | cut -d, -f2-4 | tr '\n' ' ')"
