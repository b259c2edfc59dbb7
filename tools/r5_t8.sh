#!/bin/bash
# the 2-tile 12-wave forward form of the discriminator's recurrence, stand-alone: parity and launch time
cd $GRAFT_REPO_ROOT
export RSRGAN_DFWD_T=1
timeout 900 python -m pytest tests/test_gpu_placement.py -k "persistent_discriminator and (64 or 32)" -m gpu -x -q 2>&1 | tail -4
bash tools/prof.sh t8 --steps 5 --warmup 2 --no-variants --no-kernel-timing > /dev/null 2>&1
grep -E "k_dlstm_fwd" gpurun_out/prof_t8/r_kernel_stats.csv | cut -c1-140
