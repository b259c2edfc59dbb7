#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( for i in 1 2; do for v in x0 x1; do
    echo -n "$v: "; timeout 60 tools/ubench/gpersist_trace_nt_$v 64 100 3 | head -1 | sed 's/.*slots): //'
    echo -n "$v N=32: "; timeout 60 tools/ubench/gpersist_trace_nt_$v 32 100 3 | head -1 | sed 's/.*slots): //'
  done; done ) > gpurun_out/m_harness.log 2>&1
cat gpurun_out/m_harness.log
