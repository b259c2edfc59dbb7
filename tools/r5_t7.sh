#!/bin/bash
# registers instead of scratch (X waves' stash pair, the rows' lengths): harness old / new, same-box A/B of the two libraries
cd $GRAFT_REPO_ROOT
for v in nt_old nt nt_old nt; do for m in "" b; do GP_TAGS=1 timeout 60 tools/ubench/gpersist_trace_$v 64 100 3 $m | sed "s/^/$v: /" | cut -c1-150; done; done
bash tools/ab.sh 2 tools/ab_libs/old.so tools/ab_libs/new.so
cp tools/ab_libs/new.so rsrgan_amd/lib/librsrgan_hip.so
