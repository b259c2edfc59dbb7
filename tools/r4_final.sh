#!/bin/bash
# final artefacts of a round-4 build: headline bench line, kernel stats, traffic passes, SQ / LDS counter passes, gap summary, the phase
# traces of the persistent recurrences -> gpurun_out/; then tools/mk_final.py r4 ... and copy the summaries into profiles/
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-r4fin}
timeout 400 python bench.py --steps 50 --warmup 10 > gpurun_out/bench_$tag.log 2>&1; tail -1 gpurun_out/bench_$tag.log | cut -c1-260
bash tools/prof.sh $tag --steps 5 --warmup 2 --no-variants > /dev/null 2>&1
head -16 gpurun_out/prof_$tag/r_kernel_stats.csv | cut -c1-130
bash tools/traffic.sh $tag --no-variants > gpurun_out/traffic_$tag.log 2>&1; tail -2 gpurun_out/traffic_$tag.log | cut -c1-300
bash tools/pmc.sh ${tag}_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" --steps 2 --warmup 1 --no-variants --no-kernel-timing > gpurun_out/pmc_${tag}_sq.log 2>&1; tail -14 gpurun_out/pmc_${tag}_sq.log | cut -c1-260
bash tools/pmc.sh ${tag}_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" --steps 2 --warmup 1 --no-variants --no-kernel-timing > gpurun_out/pmc_${tag}_lds.log 2>&1; tail -14 gpurun_out/pmc_${tag}_lds.log | cut -c1-260
bash tools/prof.sh ${tag}_res --net res_lstm_l --steps 5 --warmup 2 --no-variants > /dev/null 2>&1
head -10 gpurun_out/prof_${tag}_res/r_kernel_stats.csv | cut -c1-130
timeout 60 tools/ubench/gpersist_trace 64 100 3 > gpurun_out/gptrace_64.txt 2>&1; timeout 60 tools/ubench/gpersist_trace 32 50 3 > gpurun_out/gptrace_32.txt 2>&1; timeout 60 tools/ubench/gpersist_trace 64 100 3 b > gpurun_out/gptrace_64b.txt 2>&1; for m in "" b; do timeout 60 tools/ubench/gpersist_trace_nt 64 100 3 $m; timeout 60 tools/ubench/gpersist_trace_nt 32 50 3 $m; done > gpurun_out/gptrace_nt.txt 2>&1
timeout 60 tools/ubench/dpersist_trace 128 100 > gpurun_out/dptrace_fwd128.txt 2>&1; timeout 60 tools/ubench/dpersist_trace 64 100 > gpurun_out/dptrace_fwd.txt 2>&1; timeout 60 tools/ubench/dpersist_trace 128 100 bwd > gpurun_out/dptrace_bwd.txt 2>&1
head -3 gpurun_out/gptrace_64.txt
