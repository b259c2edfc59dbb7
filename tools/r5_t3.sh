#!/bin/bash
# does the busy-communication-stream failure of the trailing launch come from shared hardware queues?
cd $GRAFT_REPO_ROOT
for q in "" 8 16; do for i in 1 2 3; do
  echo -n "GPU_MAX_HW_QUEUES=$q run $i: "
  if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
  timeout 300 python -m pytest tests/test_gpu_dist.py -m gpu -x -q -k "dp_sequence_around and 64" 2>&1 | grep -E "passed|failed" | tail -1
done; done
unset GPU_MAX_HW_QUEUES
echo -n "RSRGAN_TRAIL=0: "; RSRGAN_TRAIL=0 timeout 300 python -m pytest tests/test_gpu_dist.py -m gpu -x -q -k "dp_sequence_around and 64" 2>&1 | grep -E "passed|failed" | tail -1
