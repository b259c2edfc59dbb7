#!/bin/bash
# final artefacts of the round-6 build (ONE set): GPU test suite, headline bench line (with variants), kernel stats + timeline, traffic
# passes, SQ / LDS counter passes, the shipped network's kernel stats -> gpurun_out/; then tools/mk_final.py r6 ... and copy into profiles/
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-r6fin}
(time timeout 1500 python -m pytest tests -m gpu -x -q) > gpurun_out/tests_$tag.log 2>&1; tail -4 gpurun_out/tests_$tag.log | head -2
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$tag.log 2>&1; tail -1 gpurun_out/smoke_$tag.log | cut -c1-120
timeout 1200 python bench.py --steps 50 --warmup 10 > gpurun_out/bench_$tag.log 2> gpurun_out/bench_$tag.err; tail -1 gpurun_out/bench_$tag.log | cut -c1-300
bash tools/prof.sh $tag --steps 5 --warmup 2 --no-variants > /dev/null 2>&1
f=$(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/timeline.py $f 3 > gpurun_out/timeline_$tag.txt 2>&1
head -14 gpurun_out/prof_$tag/r_kernel_stats.csv | cut -c1-130
bash tools/traffic.sh $tag --no-variants > gpurun_out/traffic_$tag.log 2>&1; tail -2 gpurun_out/traffic_$tag.log | cut -c1-300
bash tools/pmc.sh ${tag}_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" --steps 2 --warmup 1 --no-variants --no-kernel-timing > gpurun_out/pmc_${tag}_sq.log 2>&1; tail -14 gpurun_out/pmc_${tag}_sq.log | cut -c1-260
bash tools/pmc.sh ${tag}_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" --steps 2 --warmup 1 --no-variants --no-kernel-timing > gpurun_out/pmc_${tag}_lds.log 2>&1; tail -14 gpurun_out/pmc_${tag}_lds.log | cut -c1-260
bash tools/prof.sh ${tag}_res8 --net res_lstm_l --batch 8 --gen-updates 2 --steps 5 --warmup 2 --no-variants --no-kernel-timing > /dev/null 2>&1
head -8 gpurun_out/prof_${tag}_res8/r_kernel_stats.csv | cut -c1-130
