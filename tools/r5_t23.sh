#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for n in 64 32; do for sc in 0 3 4 5 6 7; do echo "N=$n sched=$sc: $(GP_TAGS=1 GP_SCHED=$sc timeout 60 tools/ubench/gpersist_trace_nt $n 100 3 | head -1 | cut -c95-150)"; done; done; done
