#!/bin/bash
cd $GRAFT_REPO_ROOT
for q in dgdgsd dgdgdgsdgsd; do echo "== $q: $(RSRGAN_DPIPE=0 SEQ=$q timeout 300 python tools/r5_dbg5.py 2>&1 | tail -1)"; done
timeout 1500 python -m pytest tests/test_gpu_placement.py -k "pipelined or trailing or persistent_generator or tagged" -m gpu -x -q 2>&1 | tail -3
echo "== whole placement + fullsize + padrows + dist under RSRGAN_DPIPE=1"
RSRGAN_DPIPE=1 timeout 2400 python -m pytest tests/test_gpu_placement.py tests/test_gpu_fullsize.py tests/test_gpu_padrows.py tests/test_gpu_dist.py -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | grep -E "^E  |FAILED|passed|failed" | head -20
