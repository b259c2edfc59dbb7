#!/bin/bash
# where do the 3 us per step of the second tile lane come from: the fabric (bytes) or the workgroup (R waves shared by the lanes)?
cd $GRAFT_REPO_ROOT
for n in 32 64; do for m in "" b; do for g in 0 1; do echo "N=$n nrt=$g: $(GP_TAGS=1 GP_NRT=$g timeout 60 tools/ubench/gpersist_trace_nt $n 100 3 $m | head -1 | cut -c1-20,95-150)"; done; done; done
