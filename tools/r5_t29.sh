#!/bin/bash
# k_glstm_bwd: the X waves' input-gradient product side by side with the R waves' state-gradient product (-DGP_NOQ) instead of behind it
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in "" _nq; do for n in 64 32; do for g in 0 1; do echo "variant=${v:-seq} N=$n nrt=$g: $(GP_TAGS=1 GP_NRT=$g timeout 60 tools/ubench/gpersist_trace_nt$v $n 100 3 b | head -1 | cut -c95-150)"; done; done; done; done
