#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_placement.py -m gpu -x -q -k "single_tile or trailing or residual" 2>&1 | tail -4
RSRGAN_DPIPE=1 timeout 900 python -m pytest tests/test_gpu_placement.py tests/test_gpu_padrows.py -m gpu -x -q -k "single_tile or padded or shipped" 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
