#!/usr/bin/env python
"""Generates tests/golden/*.npz with the fp64 CPU oracle (oracle/rsrgan_oracle.py).

The reference (Python 2 + TensorFlow 1.4) cannot run in this image and ships no fixtures, so these
vectors pin the ORACLE's output at commit time ("parity unpinned" w.r.t. the reference itself, see
the oracle header): any later change of the oracle or of the HIP path that moves a number shows up
as a diff against these files.

  small_*.npz   : a small network, everything stored (inputs, weights, losses, G(x), gradients,
                  updated weights after 1 D + 2 G updates)              -- B=4,T=7 ragged ; B=8,T=16
  reftrue_*.npz : the reference's hard-coded sizes (G lstm 3x760/p280 | res_lstm_l 4x760/p257,
                  D 2x256/p40); weights are re-generated from the seed (numpy PCG64 is stable), only
                  losses, G(x) samples and per-tensor norms are stored  -- B=4,T=6 ragged
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import rsrgan_oracle as O                                     # noqa: E402
from tests.helpers import rand_batch, rand_params, small_cfg             # noqa: E402


def run_schedule(cfg, g, d, x, lab, ln, nr, nf, **kw):
    o = O.GanRnnOracle(cfg, g, d, batch_size=x.shape[0], **kw)
    y0 = o.forward(x, ln)
    dl, dgr = o.d_tower(x.astype(np.float64), lab.astype(np.float64), ln, nr, nf)
    gl, ggr, _ = o.g_tower(x.astype(np.float64), lab.astype(np.float64), ln, nf)
    steps = [np.ravel(o.d_step(x, lab, ln, nr, nf))]
    for _ in range(2):
        steps.append(np.ravel(o.g_step(x, lab, ln, nf)))
    return o, y0, dl, dgr, gl, ggr, steps


def small(tag, g_type, B, T, ragged, seed):
    cfg = small_cfg(g_type)
    g, d = rand_params(cfg, seed)
    x, lab, ln = rand_batch(cfg, B, T, seed + 100, ragged)
    rng = np.random.default_rng(seed + 200)
    nr = rng.normal(0, 0.05, (B, 1, cfg.output_dim)); nf = rng.normal(0, 0.05, (B, 1, cfg.output_dim))
    kw = dict(l2_scale=1e-3, g_learning_rate=float(np.float32(1e-3)), d_learning_rate=float(np.float32(5e-2)))
    o, y0, dl, dgr, gl, ggr, steps = run_schedule(cfg, g, d, x, lab, ln, nr, nf, **kw)
    out = dict(g_type=g_type, x=x, lab=lab, ln=ln, noise_real=nr.astype(np.float32), noise_fake=nf.astype(np.float32),
               y0=y0, d_losses=np.array(dl), g_losses=np.array(gl), d_step=steps[0], g_step1=steps[1], g_step2=steps[2])
    for k, v in g.items(): out["g0/" + k] = v
    for k, v in d.items(): out["d0/" + k] = v
    for k, v in dgr.items(): out["dgrad/" + k] = v
    for k, v in ggr.items(): out["ggrad/" + k] = v
    for k, v in o.g.items(): out["g1/" + k] = v
    for k, v in o.d.items(): out["d1/" + k] = v
    np.savez_compressed(os.path.join(HERE, "small_%s.npz" % tag), **out)


def reftrue(tag, cfg, seed):
    B, T = 4, 6
    g, d = rand_params(cfg, seed)
    x, lab, ln = rand_batch(cfg, B, T, seed + 1, True)
    o, y0, dl, dgr, gl, ggr, steps = run_schedule(cfg, g, d, x, lab, ln, None, None)
    out = dict(seed=seed, B=B, T=T, ln=ln, y0_sample=y0[:, :, ::8], y0_abs_mean=np.abs(y0).mean(),
               d_losses=np.array(dl), g_losses=np.array(gl), d_step=steps[0], g_step1=steps[1], g_step2=steps[2])
    for k, v in dgr.items(): out["dgrad_norm/" + k] = np.linalg.norm(v)
    for k, v in ggr.items(): out["ggrad_norm/" + k] = np.linalg.norm(v)
    for k, v in o.g.items(): out["g1_delta_norm/" + k] = np.linalg.norm(v - g[k].astype(np.float64))
    np.savez_compressed(os.path.join(HERE, "reftrue_%s.npz" % tag), **out)


if __name__ == "__main__":
    small("lstm_b4t7", "lstm", 4, 7, True, 1)
    small("lstm_b8t16", "lstm", 8, 16, False, 2)
    small("res_lstm_l_b4t7", "res_lstm_l", 4, 7, True, 3)
    small("res_lstm_base_b4t7", "res_lstm_base", 4, 7, True, 4)
    reftrue("lstm", O.NetCfg(), 21)
    reftrue("res_lstm_l", O.NetCfg.res_lstm_l(), 22)
    print(sorted(os.listdir(HERE)))
