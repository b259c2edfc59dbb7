#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( for v in p0 q1 q3 p0 q1 q3; do
    for mode in "" b; do
      echo -n "$v $mode: "; timeout 60 tools/ubench/gpersist_trace_nt_$v 64 100 3 $mode | head -1 | sed 's/.*slots): //'
    done
  done ) > gpurun_out/c_harness.log 2>&1
cat gpurun_out/c_harness.log | cut -c1-200
