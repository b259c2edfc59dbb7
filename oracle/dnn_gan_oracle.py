"""CPU ORACLE (test infrastructure only) for the frame-level DNN-GAN: models/gan.py:GAN with
generator models/dnn.py:DNN and discriminator models/discriminator_dnn.py.

*** TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__ and bench.py's cpu_baseline.
*** PARITY UNPINNED w.r.t. the reference (TensorFlow 1.4 / Python 2 cannot run here; no fixtures
*** exist for this path); pinned by torch-autograd and finite differences (tests/test_oracle_dnn.py).

Restated graph (batch_norm=False, keep_prob=1.0):
  G (dnn.py:79-110)              : 1+3 = 4 x [FC 1024, ReLU], FC -> output_dim (linear)
  d_inputs (gan.py:158-160)      : inputs[:, input_dim*left_context : +input_dim]   (centre frame)
  D (discriminator_dnn.py:61-93) : concat(d_inputs, labels|G(x)) -> 4 x [FC 1024, ReLU] -> FC -> 1
                                   -> clip_by_value(-0.5, 1.5); no noise layer (:58 is commented out)
  losses (gan.py:200-214)        : LSGAN with constants 1 / 0; g_mse = 0.5*mean((g-labels)^2)*output_dim;
                                   g_l2 = sum of l2_regularizer(scale)(W) over G's FC weights (not biases)
  optimizers (gan.py:125-126)    : Adam for D and for G, NO gradient clipping, EMA 0.9999 (:128-129)
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np


@dataclass
class DnnCfg:
    input_dim: int = 257          # per frame; the fed width is input_dim*(left+1+right)
    output_dim: int = 40
    left_context: int = 5         # run_gan_dnn.sh: +-5 frames -> 2827
    right_context: int = 5
    g_units: int = 1024
    g_hidden: int = 4             # 1 + hidden_layers(3)   (dnn.py:34-35)
    d_units: int = 1024
    d_hidden: int = 4             # discriminator_dnn.py:23-24
    clip_lo: float = -0.5         # discriminator_dnn.py:93
    clip_hi: float = 1.5

    @property
    def fed_dim(self):
        return self.input_dim * (self.left_context + 1 + self.right_context)

    @property
    def joint_dim(self):
        return self.input_dim + self.output_dim


def _fc_names(prefix, n):
    return [prefix + "/fully_connected" + ("" if i == 0 else "_%d" % i) for i in range(n)]


def g_param_specs(cfg: DnnCfg) -> List[Tuple[str, Tuple[int, ...]]]:
    dims = [cfg.fed_dim] + [cfg.g_units] * cfg.g_hidden + [cfg.output_dim]
    s = []
    for i, n in enumerate(_fc_names("g_model", cfg.g_hidden + 1)):
        s += [(n + "/weights", (dims[i], dims[i + 1])), (n + "/biases", (dims[i + 1],))]
    return s


def d_param_specs(cfg: DnnCfg) -> List[Tuple[str, Tuple[int, ...]]]:
    dims = [cfg.joint_dim] + [cfg.d_units] * cfg.d_hidden + [1]
    s = []
    for i, n in enumerate(_fc_names("d_model", cfg.d_hidden + 1)):
        s += [(n + "/weights", (dims[i], dims[i + 1])), (n + "/biases", (dims[i + 1],))]
    return s


def init_params(specs, rng, dtype=np.float64, relu_init=False):
    """G: xavier_initializer() (dnn.py:85,95,106); D hidden: truncated normal std sqrt(2/units)
    (discriminator_dnn.py:25-26), D output xavier; biases zero."""
    out = {}
    last = specs[-2][0]
    for name, shape in specs:
        if name.endswith("biases"):
            out[name] = np.zeros(shape, dtype)
        elif relu_init and name != last:
            std = math.sqrt(2.0 / shape[1])
            out[name] = np.clip(rng.normal(0, std, shape), -2 * std, 2 * std).astype(dtype)
        else:
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            out[name] = rng.uniform(-lim, lim, shape).astype(dtype)
    return out


def fc_stack_fwd(P, prefix, n_layers, x):
    """n_layers FC layers, ReLU on all but the last.  Returns (y, acts) with acts[l] = input of layer l."""
    acts = [x]
    for i, n in enumerate(_fc_names(prefix, n_layers)):
        z = acts[-1] @ P[n + "/weights"] + P[n + "/biases"]
        acts.append(np.maximum(z, 0.0) if i < n_layers - 1 else z)
    return acts[-1], acts


def fc_stack_bwd(P, prefix, n_layers, acts, dy, want_dx=True):
    grads = {}
    d = dy
    names = _fc_names(prefix, n_layers)
    for i in range(n_layers - 1, -1, -1):
        if i < n_layers - 1:
            d = d * (acts[i + 1] > 0)
        grads[names[i] + "/weights"] = acts[i].T @ d
        grads[names[i] + "/biases"] = d.sum(0)
        if i > 0 or want_dx:
            d = d @ P[names[i] + "/weights"].T
    return (d if want_dx else None), grads


def d_forward(cfg, Pd, joint):
    raw, acts = fc_stack_fwd(Pd, "d_model", cfg.d_hidden + 1, joint)
    return np.clip(raw, cfg.clip_lo, cfg.clip_hi), raw, acts


def clip_grad_mask(cfg, raw):
    """tf.clip_by_value = minimum(maximum(x, lo), hi): the gradient passes where lo <= x <= hi."""
    return ((raw >= cfg.clip_lo) & (raw <= cfg.clip_hi)).astype(raw.dtype)


class GanDnnOracle:
    """models/gan.py:GAN on one tower (every tower receives the same batch there, gan.py:136)."""

    def __init__(self, cfg: DnnCfg, g, d, *, g_learning_rate=1e-4, d_learning_rate=1e-4, mse_lambda=10.0, l2_scale=0.0,
                 cross_validation=False, dtype=np.float64):
        self.cfg, self.dtype = cfg, dtype
        self.g = {k: np.array(v, dtype) for k, v in g.items()}
        self.d = {k: np.array(v, dtype) for k, v in d.items()}
        self.g_learning_rate, self.d_learning_rate = g_learning_rate, d_learning_rate
        self.mse_lambda, self.l2_scale, self.cross_validation = mse_lambda, l2_scale, cross_validation
        self.beta1, self.beta2, self.eps, self.ema_decay = 0.9, 0.999, 1e-8, 0.9999
        self.adam = {n: dict(m={k: np.zeros_like(v) for k, v in p.items()}, v={k: np.zeros_like(v) for k, v in p.items()}, t=0)
                     for n, p in (("g", self.g), ("d", self.d))}
        self.ema = {"g": {k: v.copy() for k, v in self.g.items()}, "d": {k: v.copy() for k, v in self.d.items()}}

    def _d_inputs(self, x):
        c = self.cfg
        return x[:, c.input_dim * c.left_context: c.input_dim * (c.left_context + 1)]     # gan.py:158-160

    # generator hooks (overridden by oracle/rced_oracle.py for the R-CED generator)
    def _g_fwd(self, x):
        return fc_stack_fwd(self.g, "g_model", self.cfg.g_hidden + 1, x)

    def _g_bwd(self, cache, dy):
        return fc_stack_bwd(self.g, "g_model", self.cfg.g_hidden + 1, cache, dy, want_dx=False)[1]

    def forward(self, x):
        return self._g_fwd(np.asarray(x, self.dtype))[0]

    def d_tower(self, x, lab, want_grads=True):
        cfg = self.cfg
        x, lab = np.asarray(x, self.dtype), np.asarray(lab, self.dtype)
        y = self.forward(x)
        di = self._d_inputs(x)
        losses, grads = [], None
        for joint, target in ((np.concatenate([di, lab], 1), 1.0), (np.concatenate([di, y], 1), 0.0)):
            out, raw, acts = d_forward(cfg, self.d, joint)
            diff = out - target
            losses.append(float(np.mean(diff * diff)))
            if want_grads:
                draw = 2.0 * diff / diff.size * clip_grad_mask(cfg, raw)
                _, g = fc_stack_bwd(self.d, "d_model", cfg.d_hidden + 1, acts, draw, want_dx=False)
                grads = g if grads is None else {k: grads[k] + g[k] for k in g}
        return (losses[0], losses[1], losses[0] + losses[1]), grads

    def g_tower(self, x, lab, want_grads=True):
        cfg = self.cfg
        x, lab = np.asarray(x, self.dtype), np.asarray(lab, self.dtype)
        y, gacts = self._g_fwd(x)
        supervised = getattr(self, "supervised", False)          # models/dnn_trainer.py:139-148: g_loss = g_mse + g_l2
        if supervised:
            g_adv = 0.0
        else:
            out, raw, dacts = d_forward(cfg, self.d, np.concatenate([self._d_inputs(x), y], 1))
            diff = out - 1.0
            g_adv = float(np.mean(diff * diff))
        e = y - lab
        g_mse = float(0.5 * np.mean(e * e) * cfg.output_dim)
        if (not self.cross_validation) and self.l2_scale > 0:
            g_l2 = self.l2_scale * sum(0.5 * float(np.sum(v * v)) for k, v in self.g.items() if k.endswith("weights"))
        else:
            g_l2 = 0.0
        g_loss = g_adv + self.mse_lambda * g_mse + g_l2
        grads = None
        if want_grads:
            dy = self.mse_lambda * cfg.output_dim * e / e.size
            if not supervised:
                draw = 2.0 * diff / diff.size * clip_grad_mask(cfg, raw)
                djoint, _ = fc_stack_bwd(self.d, "d_model", cfg.d_hidden + 1, dacts, draw, want_dx=True)
                dy = dy + djoint[:, cfg.input_dim:]
            grads = self._g_bwd(gacts, dy)
            if g_l2 != 0.0 or ((not self.cross_validation) and self.l2_scale > 0):
                for k in grads:
                    if k.endswith("weights"):
                        grads[k] = grads[k] + self.l2_scale * self.g[k]
        return (g_adv, g_mse, g_l2, g_loss), grads, y

    def _adam(self, which, params, grads, lr):
        st = self.adam[which]
        st["t"] += 1
        t = st["t"]
        lr_t = lr * math.sqrt(1 - self.beta2 ** t) / (1 - self.beta1 ** t)
        for k in params:
            st["m"][k] = self.beta1 * st["m"][k] + (1 - self.beta1) * grads[k]
            st["v"][k] = self.beta2 * st["v"][k] + (1 - self.beta2) * grads[k] * grads[k]
            params[k] = params[k] - lr_t * st["m"][k] / (np.sqrt(st["v"][k]) + self.eps)
            self.ema[which][k] = self.ema_decay * self.ema[which][k] + (1 - self.ema_decay) * params[k]

    def apply_d(self, grads):
        self._adam("d", self.d, grads, self.d_learning_rate)

    def apply_g(self, grads):
        self._adam("g", self.g, grads, self.g_learning_rate)

    def d_step(self, x, lab, train=True):
        losses, grads = self.d_tower(x, lab, want_grads=train)
        if train:
            self.apply_d(grads)
        return losses

    def g_step(self, x, lab, train=True):
        losses, grads, _ = self.g_tower(x, lab, want_grads=train)
        if train:
            self.apply_g(grads)
        return losses
