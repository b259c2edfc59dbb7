#!/bin/bash
cd $GRAFT_REPO_ROOT
for g in 0 1; do echo "=== nrt=$g fwd"; GP_TAGS=1 GP_NRT=$g timeout 60 tools/ubench/gpersist_trace 64 100 3 | cut -c1-200; done
for g in 0 1; do echo "=== nrt=$g bwd"; GP_TAGS=1 GP_NRT=$g timeout 60 tools/ubench/gpersist_trace 64 100 3 b | cut -c1-200; done
