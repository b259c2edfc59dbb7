#!/bin/bash
cd $GRAFT_REPO_ROOT
for k in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_placement.py -m gpu -x -q -k "single_tile" 2>&1 | tail -1; done
RSRGAN_DP_NRT=1 RSRGAN_GP_NRT=1 REP=4 timeout 600 python tools/r5_t28.py 2>&1 | tail -8 | cut -c1-200
RSRGAN_DPIPE=1 timeout 900 python -m pytest tests/test_gpu_placement.py tests/test_gpu_padrows.py -m gpu -x -q -k "single_tile or padded or shipped" 2>&1 | tail -2
