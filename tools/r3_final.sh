#!/bin/bash
# final artefacts of a build: headline bench line, kernel stats, traffic passes (cold: every dispatch instrumented; warm: one kernel
# class per pass), gap summary -> gpurun_out/, then tools/mk_final.py
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-fin}
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-variants > gpurun_out/bench_$tag.log 2>&1; tail -1 gpurun_out/bench_$tag.log | cut -c1-260
bash tools/prof.sh $tag --steps 5 --warmup 2 --no-variants > /dev/null 2>&1
head -14 gpurun_out/prof_$tag/r_kernel_stats.csv | cut -c1-130
bash tools/traffic.sh $tag --no-variants > gpurun_out/traffic_$tag.log 2>&1; tail -2 gpurun_out/traffic_$tag.log | cut -c1-300
# (tools/traffic_warm.sh: the per-class variant, a negative result kept in profiles/r3_traffic_warm_by_class.json -- not part of the final set)
# variants: R-CED GAN and SEGAN kernel statistics
bash tools/prof.sh rced --net rced --rced-gan --batch 6400 --rced-width 257 --steps 1 --warmup 1 --no-variants > /dev/null 2>&1
head -8 gpurun_out/prof_rced/r_kernel_stats.csv | cut -c1-130
bash tools/prof.sh seg --net segan --batch 32 --steps 3 --warmup 1 --no-variants > /dev/null 2>&1
head -8 gpurun_out/prof_seg/r_kernel_stats.csv | cut -c1-130
timeout 60 tools/ubench/dpersist_trace 64 100 > gpurun_out/dptrace_fwd.txt 2>&1; timeout 60 tools/ubench/dpersist_trace 128 100 bwd > gpurun_out/dptrace_bwd.txt 2>&1
