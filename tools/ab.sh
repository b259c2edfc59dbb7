#!/bin/bash
# same-box A/B of two builds of the library: tools/ab.sh <rounds> <libA.so> <libB.so> [bench args...]
cd $GRAFT_REPO_ROOT
n=$1; A=$2; B=$3; shift 3
for i in $(seq $n); do
  for L in $A $B; do
    cp $L rsrgan_amd/lib/librsrgan_hip.so
    timeout 300 python bench.py --steps 40 --warmup 10 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing "$@" > gpurun_out/ab.log 2>&1
    echo "$(basename $L): $(tail -1 gpurun_out/ab.log | grep -o '"ms_per_step": [0-9.]*, "ms_per_step_median": [0-9.]*')"
  done
done
