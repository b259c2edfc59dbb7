#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python - <<'PY' > gpurun_out/j_mask.log 2>&1
import json, os, subprocess, sys, time
sys.path.insert(0, ".")
from tests.test_gpu_padrows import WORKER
def run(env):
    e = dict(os.environ); e.update(env)
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c", WORKER], capture_output=True, text=True, env=e, timeout=900)
    dt = time.time() - t0
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    r = json.loads(line[-1][7:]) if line else {"err": p.stderr[-500:]}
    print(env, "%.1fs" % dt, {k: r.get(k) for k in ("status", "n_gp", "ok", "cus", "err")}, [round(v[0], 4) for v in r.get("d", [])])
run({"RSRGAN_TEST_FLAGS": "1"})
run({"HSA_CU_MASK": "0:0-127", "RSRGAN_TEST_FLAGS": "1"})
run({"HSA_CU_MASK": "0:0-127", "RSRGAN_TEST_FLAGS": "1", "RSRGAN_RESIDENT_PROBE": "0"})
run({"HSA_CU_MASK": "0:0-127", "RSRGAN_TEST_FLAGS": "3", "RSRGAN_RESIDENT_PROBE": "0"})
run({"ROC_GLOBAL_CU_MASK": "0x" + "f" * 32, "RSRGAN_TEST_FLAGS": "1", "RSRGAN_RESIDENT_PROBE": "0"})
PY
cat gpurun_out/j_mask.log
