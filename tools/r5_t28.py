"""flaky-bit hunt: the placement worker at B=8, T=1 several times; prints what differs between runs"""
import os, sys, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_placement import _run
size = {"RSRGAN_TEST_B": "8", "RSRGAN_TEST_T": os.environ.get("TT", "1"), "RSRGAN_TEST_NET": "lstm", "RSRGAN_PAD_ROWS": "1", "RSRGAN_TEST_REUSE": "1"}
ref = None
for k in range(int(os.environ.get("REP", "6"))):
    for gp in ("1", "0"):
        a = _run(dict(size, RSRGAN_GP_NRT=gp))
        key = {x: a[x] for x in ("d0", "g0", "d1", "g1")}
        if ref is None: ref = key
        diff = [x for x in key if key[x] != ref[x]]
        print("run", k, "gp_nrt", gp, "status", a["device_status"], "sha", a["vars_sha"][:10], "differs:", {x: (key[x], ref[x]) for x in diff})
