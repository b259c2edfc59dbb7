#!/bin/bash
# usage: tools/prof.sh <tag> <bench args...> : rocprofv3 kernel trace -> gpurun_out/prof_<tag>/ (csv stats only)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r -- python $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline --no-hbm-activity > $out/bench.log 2>&1
cd $GRAFT_REPO_ROOT
find $out -name "*kernel_trace*" -size +20M -delete
ls $out | head
