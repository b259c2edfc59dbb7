#!/bin/bash
# round 6, last call: the GPU suite as the driver runs it (time), smoke, and the bench line with profiles/r6_final_* in place
cd $GRAFT_REPO_ROOT
(time timeout 1500 python -m pytest tests -m gpu -x -q --durations=8) > gpurun_out/tests_r6k.log 2>&1; grep -E "passed|failed" gpurun_out/tests_r6k.log; grep real gpurun_out/tests_r6k.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_r6k.log 2>&1; tail -1 gpurun_out/smoke_r6k.log | cut -c1-100
(time timeout 1200 python bench.py --steps 50 --warmup 10) > gpurun_out/bench_r6k.log 2> gpurun_out/bench_r6k.err; tail -1 gpurun_out/bench_r6k.log | cut -c1-260; grep real gpurun_out/bench_r6k.err
