import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rsrgan_oracle as O
from tests.helpers import build_hip_pair, rand_batch
import torch
cfg = O.NetCfg()
B, T = 64, 100
model, _ = build_hip_pair(cfg, B, T, seed=5, flags=3)
x, lab, ln = rand_batch(cfg, B, T, seed=6, ragged=True)
for c in os.environ["SEQ"]:
    if c == "d": print("d", np.ravel(model.d_step(x, lab, ln)), flush=True)
    elif c == "g": print("g", np.ravel(model.g_step(x, lab, ln, reuse_g_forward=True)), flush=True)
    elif c == "s": print("status", model.engine.device_status(), flush=True)
    elif c == "y": torch.cuda.synchronize(); print("sync", flush=True)
print("final status", model.engine.device_status())
