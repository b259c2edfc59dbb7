#!/bin/bash
cd $GRAFT_REPO_ROOT
for g in 1 0; do echo "== NRT=$g"; RSRGAN_GP_NRT=$g RSRGAN_DPIPE=1 timeout 600 python -m pytest tests/test_gpu_padrows.py -m gpu -x -q -k "shipped_recipe_batch_against_oracle and 8-12-1" 2>&1 | grep -E "^E  |passed|failed" | head -12; done
