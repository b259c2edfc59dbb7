"""Golden vector for BASELINE.json configs[4] at its REAL batch: the SEGAN-style conv G/D (models/segan.py:155-236, generator.py:112-295,
discriminator.py:20-95, utils/bnorm.py:11-69) on 16384-sample chunks with B = 32, computed by the fp64 oracle (oracle/segan_oracle.py)
in the build container.  At B = 32 the virtual batch norm mixes with 1 / (B + 1) = 1/33 (bnorm.py:36-48) and the HIP path's GEMM
planner takes other branches (M = 32 L) than in the B = 2 case of tests/test_gpu_segan.py -- and B = 32 is what bench.py times.
Inputs are NOT stored: tests/test_gpu_segan.py rebuilds them from the same seeds with the same helper (_pair / _batch); stored are
the towers' losses of the D-run and the G-run, a sample of G(x), the norm of every gradient tensor, and -- after ONE RMSProp step
of each net -- the losses of the next fetch and the norm of every variable's change.
Run:  python tests/golden/make_segan_golden.py      (about ten minutes of CPU, ~20 GB)"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import segan_oracle as S                      # noqa: E402

SEED, B, L, U = 32, 32, 16384, 40


def build(seed=SEED, g_lr=2e-4, d_lr=3e-4, l1=100.0):
    """the oracle half of tests/test_gpu_segan.py::_pair and ::_batch (same draws in the same order)"""
    cfg = S.SeganCfg(input_len=L, output_dim=U, g_depths=tuple(S.DEPTHS), d_depths=tuple(S.DEPTHS), g_kwidth=20, d_kwidth=31, g_nl="prelu")
    rng = np.random.default_rng(seed)
    g = S.init_params(S.g_param_specs(cfg), rng); d = S.init_params(S.d_param_specs(cfg), rng)
    for p in (g, d):
        for k in p:
            if not (k.endswith("/W") or k.endswith("kernel") or k.endswith("weights")):
                p[k] = p[k] + 0.1 * rng.standard_normal(p[k].shape)
            p[k] = p[k].astype(np.float32)
    o = S.SeganOracle(cfg, g, d, batch_size=B, g_learning_rate=float(np.float32(g_lr)), d_learning_rate=float(np.float32(d_lr)), l1_lambda=l1)
    n = len(cfg.g_depths)
    x = rng.standard_normal((B, cfg.input_len)).astype(np.float32); lab = rng.standard_normal((B, cfg.output_dim)).astype(np.float32)
    z = rng.standard_normal((B, S.enc_lengths(cfg.input_len, n)[-1], cfg.g_depths[-1])).astype(np.float32)
    nz = [(0.1 * rng.standard_normal((B, cfg.input_len + cfg.output_dim))).astype(np.float32) for _ in range(3)]
    return cfg, g, d, o, x, lab, z, nz


def main():
    t0 = time.time()
    cfg, g, d, o, x, lab, z, nz = build()
    y = o.forward(x, z)
    out = dict(seed=SEED, B=B, L=L, U=U, y_sample=y[:, ::4].astype(np.float64), y_abs_mean=np.abs(y).mean())
    dl, dg = o.d_tower(x, lab, z, *nz)
    gl, gg, _ = o.g_tower(x, lab, z, nz[0], nz[2])
    out["d_losses"] = np.asarray(dl, np.float64); out["g_losses"] = np.asarray(gl, np.float64)
    for k, v in dg.items(): out["dgrad_norm/" + k] = np.linalg.norm(v)
    for k, v in gg.items(): out["ggrad_norm/" + k] = np.linalg.norm(v)
    print("towers done", round(time.time() - t0), "s", flush=True)
    out["d_step"] = np.asarray(o.d_step(x, lab, z, *nz), np.float64)           # one RMSProp step of D, then of G (on the updated D)
    out["g_step"] = np.asarray(o.g_step(x, lab, z, nz[0], nz[2]), np.float64)
    for k, v in o.d.items(): out["d1_delta_norm/" + k] = np.linalg.norm(v.astype(np.float64) - d[k].astype(np.float64))
    for k, v in o.g.items(): out["g1_delta_norm/" + k] = np.linalg.norm(v.astype(np.float64) - g[k].astype(np.float64))
    out["d_next"] = np.asarray(o.d_tower(x, lab, z, *nz)[0], np.float64)       # the losses of the updated nets
    np.savez_compressed(os.path.join(HERE, "segan_b32_l16384.npz"), **out)
    print("written", os.path.join(HERE, "segan_b32_l16384.npz"), round(time.time() - t0), "s")


if __name__ == "__main__":
    main()
