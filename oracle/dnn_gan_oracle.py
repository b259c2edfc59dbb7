"""CPU ORACLE (test infrastructure only) for the frame-level DNN-GAN: models/gan.py:GAN with
generator models/dnn.py:DNN and discriminator models/discriminator_dnn.py.

*** TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__ and bench.py's cpu_baseline.
*** PARITY UNPINNED w.r.t. the reference (TensorFlow 1.4 / Python 2 cannot run here; no fixtures
*** exist for this path); pinned by torch-autograd and finite differences (tests/test_oracle_dnn.py).

`DnnCfg.batch_norm=True` (run_gan_dnn.sh:134, run_dnn.sh:134): every hidden fully_connected becomes
relu(batch_norm(x.W)) without a bias (contrib fully_connected drops `biases` when a normalizer_fn is given); the
normaliser and the order of its state updates are restated in oracle/bn_renorm.py.

`keep_prob < 1` (dnn.py:86,99,116-121; discriminator_dnn.py:68,81,100-105): tf.nn.dropout after every hidden ReLU of both
nets, y = x / keep_prob * mask (TF 1.4 nn_ops.dropout: `math_ops.div(x, keep_prob) * binary_tensor`), active only when
l2_scale > 0 and is_training (dnn.py:67-71 / discriminator_dnn.py:47-51 reset keep_prob to 1.0 otherwise).  TF's random stream
cannot be reproduced, so the MASKS ARE AN INPUT of this oracle: `mask_fn(run, net, layer, call, rows, cols)` -> {0,1} array, with
run = index of the training sess.run (1, 2, ...), net 0 = G / 1 = D, call 0 = D on the real joint / 1 = D on the fake joint.

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

from . import bn_renorm as bn


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
    batch_norm: bool = False      # normalizer_fn=batch_norm(scale=True, renorm=True) on the hidden layers (dnn.py:56-61)

    @property
    def fed_dim(self):
        return self.input_dim * (self.left_context + 1 + self.right_context)

    @property
    def joint_dim(self):
        return self.input_dim + self.output_dim


def _fc_names(prefix, n):
    return [prefix + "/fully_connected" + ("" if i == 0 else "_%d" % i) for i in range(n)]


def _fc_specs(prefix, dims, batch_norm):
    s = []
    names = _fc_names(prefix, len(dims) - 1)
    for i, n in enumerate(names):
        s.append((n + "/weights", (dims[i], dims[i + 1])))
        if batch_norm and i < len(names) - 1:
            s += bn.var_specs(n, dims[i + 1])
        else:
            s.append((n + "/biases", (dims[i + 1],)))
    return s


def g_param_specs(cfg: DnnCfg) -> List[Tuple[str, Tuple[int, ...]]]:
    return _fc_specs("g_model", [cfg.fed_dim] + [cfg.g_units] * cfg.g_hidden + [cfg.output_dim], cfg.batch_norm)


def d_param_specs(cfg: DnnCfg) -> List[Tuple[str, Tuple[int, ...]]]:
    return _fc_specs("d_model", [cfg.joint_dim] + [cfg.d_units] * cfg.d_hidden + [1], cfg.batch_norm)


def init_params(specs, rng, dtype=np.float64, relu_init=False):
    """G: xavier_initializer() (dnn.py:85,95,106); D hidden: truncated normal std sqrt(2/units)
    (discriminator_dnn.py:25-26), D output xavier; biases zero; BatchNorm variables as normalization.py builds them."""
    out = {}
    last = [n for n, _ in specs if n.endswith("/weights")][-1]
    for name, shape in specs:
        if "/BatchNorm/" in name:
            one = name.endswith("gamma") or name.endswith("moving_variance")
            out[name] = (np.ones if one else np.zeros)(shape, dtype)
        elif name.endswith("biases"):
            out[name] = np.zeros(shape, dtype)
        elif relu_init and name != last:
            std = math.sqrt(2.0 / shape[1])
            out[name] = np.clip(rng.normal(0, std, shape), -2 * std, 2 * std).astype(dtype)
        else:
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            out[name] = rng.uniform(-lim, lim, shape).astype(dtype)
    return out


def trainable(name):
    return not bn.is_state(name)


def fc_stack_fwd(P, prefix, n_layers, x, training=True, drop=None):
    """n_layers FC layers, ReLU on all but the last.  Returns (y, acts) with acts[l] = input of layer l; a hidden layer with
    `<name>/BatchNorm/*` variables is relu(batch_norm(x.W)) and acts carries its cache in acts.bn[l].
    drop = (keep_prob, mask_of_layer(layer, rows, cols)) or None: tf.nn.dropout on every hidden output; acts.drop[l] keeps
    (relu output, mask) of layer l."""
    acts = _Acts([x])
    for i, n in enumerate(_fc_names(prefix, n_layers)):
        if n + "/BatchNorm/beta" in P:
            z = acts[-1] @ P[n + "/weights"]
            if training:
                z, cache = bn.forward_train(P, n, z)
                acts.bn[i] = cache
            else:
                z = bn.forward_infer(P, n, z)
        else:
            z = acts[-1] @ P[n + "/weights"] + P[n + "/biases"]
        if i < n_layers - 1:
            h = np.maximum(z, 0.0)
            if drop is not None:
                keep, mask_of = drop
                m = np.asarray(mask_of(i, h.shape[0], h.shape[1]), h.dtype)
                acts.drop[i] = (h, m, keep)
                h = h / keep * m
            acts.append(h)
        else:
            acts.append(z)
    return acts[-1], acts


class _Acts(list):
    def __init__(self, it):
        super().__init__(it)
        self.bn = {}
        self.drop = {}


def fc_stack_bwd(P, prefix, n_layers, acts, dy, want_dx=True):
    grads = {}
    d = dy
    names = _fc_names(prefix, n_layers)
    for i in range(n_layers - 1, -1, -1):
        if i < n_layers - 1:
            dr = getattr(acts, "drop", {}).get(i)
            if dr is not None:                       # gradient of div(x, keep) * mask, then the ReLU on its own output
                d = d * dr[1] / dr[2] * (dr[0] > 0)
            else:
                d = d * (acts[i + 1] > 0)
        cache = getattr(acts, "bn", {}).get(i)
        if cache is not None:
            d, gb = bn.backward_train(P, cache, d)
            grads.update(gb)
        elif names[i] + "/biases" in P:
            grads[names[i] + "/biases"] = d.sum(0)
        grads[names[i] + "/weights"] = acts[i].T @ d
        if i > 0 or want_dx:
            d = d @ P[names[i] + "/weights"].T
    return (d if want_dx else None), grads


def bn_commit(P, acts, times=1):
    """The batch-norm UPDATE_OPS of one forward call, `times` times (the graph holds that many identical calls)."""
    for _ in range(times):
        for i in sorted(getattr(acts, "bn", {})):
            bn.commit(P, acts.bn[i])


def d_forward(cfg, Pd, joint, training=True, drop=None):
    raw, acts = fc_stack_fwd(Pd, "d_model", cfg.d_hidden + 1, joint, training, drop)
    return np.clip(raw, cfg.clip_lo, cfg.clip_hi), raw, acts


def clip_grad_mask(cfg, raw):
    """tf.clip_by_value = minimum(maximum(x, lo), hi): the gradient passes where lo <= x <= hi."""
    return ((raw >= cfg.clip_lo) & (raw <= cfg.clip_hi)).astype(raw.dtype)


class GanDnnOracle:
    """models/gan.py:GAN on one tower (every tower receives the same batch there, gan.py:136)."""

    def __init__(self, cfg: DnnCfg, g, d, *, g_learning_rate=1e-4, d_learning_rate=1e-4, mse_lambda=10.0, l2_scale=0.0,
                 cross_validation=False, dtype=np.float64, keep_prob=1.0, mask_fn=None):
        self.cfg, self.dtype = cfg, dtype
        self.keep_prob, self.mask_fn, self._run = keep_prob, mask_fn, 0
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

    _eval_call = False

    @property
    def training(self):
        """is_training of the batch-norm layers: False on the cross_validation twin (dnn.py:49-50, discriminator_dnn.py:29).
        d_step / g_step with train=False ARE that twin's fetches on the shared variables (train_gan_dnn.py:182-215 runs them on
        cv_model), so they normalise with the moving statistics too and carry no L2 term (gan.py:207)."""
        return not self.cross_validation and not self._eval_call

    def _drop(self, net, call=0):
        """(keep_prob, mask_of_layer) of one forward call of this run, or None: dropout only acts when l2_scale > 0 and
        is_training (dnn.py:67-71, discriminator_dnn.py:47-51)."""
        if not (self.keep_prob < 1.0 and self.l2_scale > 0 and not self.cross_validation and not self._eval_call):
            return None
        run = self._run
        return self.keep_prob, (lambda layer, rows, cols: self.mask_fn(run, net, layer, call, rows, cols))

    # generator hooks (overridden by oracle/rced_oracle.py for the R-CED generator)
    def _g_fwd(self, x):
        return fc_stack_fwd(self.g, "g_model", self.cfg.g_hidden + 1, x, self.training, self._drop(0))

    def _g_bwd(self, cache, dy):
        return fc_stack_bwd(self.g, "g_model", self.cfg.g_hidden + 1, cache, dy, want_dx=False)[1]

    def _g_commit(self, cache, times):
        bn_commit(self.g, cache, times)

    def forward(self, x):
        return self._g_fwd(np.asarray(x, self.dtype))[0]

    def _commit_run(self, gacts, d_real_acts, d_fake_acts):
        """Every batch-norm update op of the graph runs in every training `sess.run` (gan.py:139-146); tower 0 holds two
        generator calls, a dummy + a real discriminator call on the real joint and one on the fake joint (gan.py:162-181)."""
        if not self.training:
            return
        self._g_commit(gacts, 2)
        if d_real_acts is not None:
            bn_commit(self.d, d_real_acts, 2)
        if d_fake_acts is not None:
            bn_commit(self.d, d_fake_acts, 1)

    def d_tower(self, x, lab, want_grads=True):
        cfg = self.cfg
        self._run += 1 if want_grads else 0                      # every training sess.run draws new dropout masks
        x, lab = np.asarray(x, self.dtype), np.asarray(lab, self.dtype)
        y, gacts = self._g_fwd(x)
        di = self._d_inputs(x)
        losses, grads, dacts = [], None, []
        for call, (joint, target) in enumerate(((np.concatenate([di, lab], 1), 1.0), (np.concatenate([di, y], 1), 0.0))):
            out, raw, acts = d_forward(cfg, self.d, joint, self.training, self._drop(1, call))
            dacts.append(acts)
            diff = out - target
            losses.append(float(np.mean(diff * diff)))
            if want_grads:
                draw = 2.0 * diff / diff.size * clip_grad_mask(cfg, raw)
                _, g = fc_stack_bwd(self.d, "d_model", cfg.d_hidden + 1, acts, draw, want_dx=False)
                grads = g if grads is None else {k: grads[k] + g[k] for k in g}
        if want_grads:
            self._commit_run(gacts, dacts[0], dacts[1])
        return (losses[0], losses[1], losses[0] + losses[1]), grads

    def g_tower(self, x, lab, want_grads=True):
        cfg = self.cfg
        self._run += 1 if want_grads else 0
        x, lab = np.asarray(x, self.dtype), np.asarray(lab, self.dtype)
        y, gacts = self._g_fwd(x)
        supervised = getattr(self, "supervised", False)          # models/dnn_trainer.py:139-148: g_loss = g_mse + g_l2
        d_real_acts = dacts = None
        if supervised:
            g_adv = 0.0
        else:
            out, raw, dacts = d_forward(cfg, self.d, np.concatenate([self._d_inputs(x), y], 1), self.training, self._drop(1, 1))
            diff = out - 1.0
            g_adv = float(np.mean(diff * diff))
            if want_grads and self.training and cfg.batch_norm:      # the real-joint call only contributes its update ops here
                d_real_acts = d_forward(cfg, self.d, np.concatenate([self._d_inputs(x), lab], 1), True, self._drop(1, 0))[2]
        e = y - lab
        g_mse = float(0.5 * np.mean(e * e) * cfg.output_dim)
        if self.training and self.l2_scale > 0:
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
            if g_l2 != 0.0 or (self.training and self.l2_scale > 0):
                for k in grads:
                    if k.endswith("weights"):
                        grads[k] = grads[k] + self.l2_scale * self.g[k]
            self._commit_run(gacts, d_real_acts, dacts)
        return (g_adv, g_mse, g_l2, g_loss), grads, y

    def _adam(self, which, params, grads, lr):
        st = self.adam[which]
        st["t"] += 1
        t = st["t"]
        lr_t = lr * math.sqrt(1 - self.beta2 ** t) / (1 - self.beta1 ** t)
        for k in params:
            if not trainable(k):                 # batch-norm statistics: not in tf.trainable_variables()
                continue
            st["m"][k] = self.beta1 * st["m"][k] + (1 - self.beta1) * grads[k]
            st["v"][k] = self.beta2 * st["v"][k] + (1 - self.beta2) * grads[k] * grads[k]
            params[k] = params[k] - lr_t * st["m"][k] / (np.sqrt(st["v"][k]) + self.eps)
            self.ema[which][k] = self.ema_decay * self.ema[which][k] + (1 - self.ema_decay) * params[k]

    def apply_d(self, grads):
        self._adam("d", self.d, grads, self.d_learning_rate)

    def apply_g(self, grads):
        self._adam("g", self.g, grads, self.g_learning_rate)

    def d_step(self, x, lab, train=True):
        self._eval_call = not train
        try:
            losses, grads = self.d_tower(x, lab, want_grads=train)
        finally:
            self._eval_call = False
        if train:
            self.apply_d(grads)
        return losses

    def g_step(self, x, lab, train=True):
        self._eval_call = not train
        try:
            losses, grads, _ = self.g_tower(x, lab, want_grads=train)
        finally:
            self._eval_call = False
        if train:
            self.apply_g(grads)
        return losses
