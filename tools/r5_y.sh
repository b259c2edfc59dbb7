#!/bin/bash
# experiment: what the re-arming stores of the ring slots cost the persistent generator launches (rings as deep as the launch, armed by memsets outside the timed region)
cd $GRAFT_REPO_ROOT
for v in nt nr nt nr; do
for m in "" b; do for n in 64 32; do timeout 60 tools/ubench/gpersist_trace_$v $n 100 3 $m | sed "s/^/$v: /" | cut -c1-150; done; done
done
