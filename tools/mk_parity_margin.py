#!/usr/bin/env python
"""Fold the per-case records tests/test_gpu_fullsize.py::test_full_size_step_as_benched_against_oracle wrote on the GPU box
(gpurun_out/parity_margin/<net>_B<b>_T<t>_dpipe<d>_tags<g>.json: ACHIEVED error of the HIP path against the fp64 oracle) into
profiles/r6_parity_margin.json, which bench.py quotes in its JSON line ("parity_margin").  Usage: python tools/mk_parity_margin.py [note]"""
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cases = {}
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "parity_margin", "*.json"))):
    r = json.load(open(f))
    cases[os.path.basename(f)[:-5]] = {k: r[k] for k in ("loss", "loss_after_updates", "grad_d", "grad_g", "mfcc_l1", "worst", "device_status",
                                                         "standalone_g_forward_launches", "case")}
if not cases:
    sys.exit("no records under gpurun_out/parity_margin/")
doc = {"what": "max relative error of the 7 losses (before / after three updates), worst gradient tensor's relative L2 error per net, "
               "enhanced-MFCC mean-L1 relative error: librsrgan_hip.so on an MI355X against oracle/rsrgan_oracle.py in fp64, three ragged "
               "batches in turn, steps enqueued without a host wait, hipGraph replay (flags 3)",
       "bounds": {"loss": 1e-3, "grad": 2e-3, "mfcc": 1e-3},
       "measured": (sys.argv[1] if len(sys.argv) > 1 else time.strftime("%Y-%m-%d")), "cases": cases}
out = os.path.join(ROOT, "profiles", "r6_parity_margin.json")
json.dump(doc, open(out, "w"), indent=1, sort_keys=True)
print(out, len(cases), "cases")
