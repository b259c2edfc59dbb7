#!/bin/bash
# sched bit 2: the R waves' cell phase at priority 2 (it ran at 0, below the X waves' bursts at 1)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for m in "" b; do for n in 64 32; do for sc in 0 3 4 7; do echo "N=$n ${m:-f} sched=$sc: $(GP_TAGS=1 GP_SCHED=$sc timeout 60 tools/ubench/gpersist_trace_nt $n 100 3 $m | head -1 | cut -c95-150)"; done; done; done; done
