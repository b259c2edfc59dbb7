#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_e
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_e -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 6 --no-variants --no-cpu-baseline --no-hbm-activity --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/prof_e/bench.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_e -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f 2 > gpurun_out/e_timeline.txt 2>&1
find gpurun_out/prof_e -name "*kernel_trace*" -delete
tail -80 gpurun_out/e_timeline.txt
