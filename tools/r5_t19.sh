#!/bin/bash
# two tile lanes cost 3-4 us per step over one: which of the workgroup's other activities lengthens a lane's hand-off?  (forward, N=64, 3 layers)
cd $GRAFT_REPO_ROOT
for v in "" _a1 _a2 _a3 _a4 _a7; do for g in 0 1; do echo "variant=${v:-base} nrt=$g: $(GP_TAGS=1 GP_NRT=$g timeout 60 tools/ubench/gpersist_trace_nt$v 64 100 3 | head -1 | cut -c95-150)"; done; done
