#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export GP_LOOP_SECONDS=4
( for v in s0 s1; do
    tools/hbm_phase.sh "k_glstm_fwd $v" tools/ubench/gpersist_trace_nt_$v 64 100 3
    tools/hbm_phase.sh "k_glstm_bwd $v" tools/ubench/gpersist_trace_nt_$v 64 100 3 b
  done
  tools/hbm_phase.sh "k_glstm_fwd s0 N=32" tools/ubench/gpersist_trace_nt_s0 32 100 3
) > gpurun_out/g_hbm.log 2>&1
cat gpurun_out/g_hbm.log
