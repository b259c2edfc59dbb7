"""Debug helper: G gradients of one step with the persistent generator BPTT (RSRGAN_GPERSIST=3) against the launch-per-phase
backward (RSRGAN_GPERSIST=1), per tensor.  Usage: python tools/gp_bwd_check.py [B] [T]   (spawns two worker processes)."""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r"""
import sys, os, numpy as np
sys.path.insert(0, %r)
from oracle import rsrgan_oracle as O
from tests.helpers import NET_D, NET_G, build_hip_pair, rand_batch
cfg = O.NetCfg()
B, T = int(os.environ["GB"]), int(os.environ["GT"])
model, _ = build_hip_pair(cfg, B, T, seed=5, flags=int(os.environ.get("GFLAGS", "3")))
x, lab, ln = rand_batch(cfg, B, T, seed=6, ragged=os.environ.get("GRAGGED", "1") == "1")
E = model.engine
d = E.d_backward(x, lab, ln, train=True, apply=False).cpu().numpy()
g = E.g_backward(x, lab, ln, train=True, reuse=True, apply=False).cpu().numpy()
flat = E.get_grads(NET_G).cpu().numpy()
gg = {}
for name, shape, off in E.tensor_table(NET_G):
    gg[name.replace("/", "__")] = flat[off:off + int(np.prod(shape))]
import time
t0 = time.time()
for _ in range(5): model.d_step(x, lab, ln); model.g_step(x, lab, ln, reuse_g_forward=True)
import torch; torch.cuda.synchronize()
print("status", model.engine.device_status(), "losses", np.ravel(d), np.ravel(g), "ms/step", (time.time() - t0) / 5 * 1e3)
np.savez(os.environ["GOUT"], **gg)
""" % ROOT

def run(env, out):
    e = dict(os.environ); e.update(env); e["GOUT"] = out
    p = subprocess.run([sys.executable, "-c", WORKER], capture_output=True, text=True, env=e, timeout=900)
    print(p.stdout[-1500:]); 
    if p.returncode: print(p.stderr[-3000:]); sys.exit(1)
    return dict(np.load(out))

B = sys.argv[1] if len(sys.argv) > 1 else "32"; T = sys.argv[2] if len(sys.argv) > 2 else "9"
base = {"GB": B, "GT": T}
a = run(dict(base, RSRGAN_GPERSIST="3"), "/tmp/gpb_a.npz")
b = run(dict(base, RSRGAN_GPERSIST="1"), "/tmp/gpb_b.npz")
worst = 0.0
for k in sorted(a):
    da = a[k].astype(np.float64); db = b[k].astype(np.float64)
    rel = np.linalg.norm(da - db) / max(np.linalg.norm(db), 1e-30)
    worst = max(worst, rel)
    print("%-60s |ref| %.4e  rel diff %.3e %s" % (k, np.linalg.norm(db), rel, "  <<<" if not rel < 1e-4 else ""))
print("worst", worst)
