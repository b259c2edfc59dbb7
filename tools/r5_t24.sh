#!/bin/bash
# the backward kernel's off-chain work with one and two lanes (timing ablations -DGP_ABL2: 8 no input-gradient products, 16 no dz stores, 32 no stash prefetch)
cd $GRAFT_REPO_ROOT
for v in "" _b1 _b2 _b3 _b7; do for g in 0 1; do echo "variant=${v:-base} nrt=$g: $(GP_TAGS=1 GP_NRT=$g timeout 60 tools/ubench/gpersist_trace_nt$v 64 100 3 b | head -1 | cut -c95-150)"; done; done
