"""CPU ORACLE twin in PyTorch-CPU autograd (test infrastructure only).

*** TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
*** bench.py's cpu_baseline leg.  Never by rsrgan_amd/.
*** PARITY UNPINNED (see oracle/rsrgan_oracle.py header).

Same restatement as oracle/rsrgan_oracle.py, but the backward pass comes from
torch.autograd instead of hand-written BPTT, so the two check each other; and
because ATen/oneDNN GEMMs use all host cores this twin is the CPU baseline
that bench.py times beside the MI355X path.
Reference lines restated: models/gan_rnn_placeholder.py:139-298,
models/lstm.py:41-129, models/res_lstm_l.py:41-199,
models/discriminator_lstm.py:24-110, utils/ops.py:19-30,120-121.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

from .rsrgan_oracle import NetCfg, g_param_specs, d_param_specs


def _lstmp(x, lengths, K, b, wf, wi, wo, Wp, forget_bias):
    """dynamic_rnn(LSTMCell(peepholes, num_proj)); x [B,T,I] (BNLSTMCell.py:176-217)."""
    B, T, _ = x.shape
    H = wf.shape[0]
    R = Wp.shape[1] if Wp is not None else H
    c = x.new_zeros(B, H)
    m = x.new_zeros(B, R)
    outs = []
    for t in range(T):
        mask = (lengths > t).unsqueeze(1)
        z = torch.cat([x[:, t], m], 1) @ K + b
        i, j, f, o = z.split(H, dim=1)
        cn = torch.sigmoid(f + forget_bias + wf * c) * c + torch.sigmoid(i + wi * c) * torch.tanh(j)
        h = torch.sigmoid(o + wo * cn) * torch.tanh(cn)
        mn = h @ Wp if Wp is not None else h
        outs.append(torch.where(mask, mn, torch.zeros_like(mn)))
        c = torch.where(mask, cn, c)
        m = torch.where(mask, mn, m)
    return torch.stack(outs, 1)


def _layer(P, pre, has_proj):
    return (P[pre + "/kernel"], P[pre + "/bias"], P[pre + "/w_f_diag"], P[pre + "/w_i_diag"],
            P[pre + "/w_o_diag"], P[pre + "/projection/kernel"] if has_proj else None)


def generator(cfg: NetCfg, P, x, lengths):
    hp = cfg.g_proj > 0
    if cfg.g_type == "lstm":
        a = x @ P["g_model/fully_connected/weights"] + P["g_model/fully_connected/biases"]
        h = torch.maximum(a, cfg.lrelu_alpha * a)
        for l in range(cfg.g_layers):
            h = _lstmp(h, lengths, *_layer(P, "g_model/rnn/multi_rnn_cell/cell_%d/lstm_cell" % l, hp), cfg.forget_bias)
        return h @ P["g_model/fully_connected_1/weights"] + P["g_model/fully_connected_1/biases"]
    res = cfg.g_type == "res_lstm_l"
    h = x
    for l in range(cfg.g_layers):
        o = _lstmp(h, lengths, *_layer(P, "g_model/lstm_cell_%d/rnn/lstm_cell" % (l + 1), hp), cfg.forget_bias)
        h = o + h if res else o
    return h @ P["g_model/forward_out/fully_connected/weights"] + P["g_model/forward_out/fully_connected/biases"]


def discriminator(cfg: NetCfg, P, x, lengths, noise=None):
    if cfg.d_type == "dnn":                      # models/discriminator_dnn.py:61-93 (no noise layer)
        h = x
        for i in range(cfg.d_layers + 1):
            n = "d_model/fully_connected" + ("" if i == 0 else "_%d" % i)
            h = h @ P[n + "/weights"] + P[n + "/biases"]
            if i < cfg.d_layers:
                h = torch.relu(h)
        return torch.clamp(h, -0.5, 1.5)
    hp = cfg.d_proj > 0
    h = x if noise is None else x + noise
    for l in range(cfg.d_layers):
        h = _lstmp(h, lengths, *_layer(P, "d_model/rnn/multi_rnn_cell/cell_%d/lstm_cell" % l, hp), cfg.forget_bias)
    return h @ P["d_model/fully_connected/weights"] + P["d_model/fully_connected/biases"]


def _clip_by_norm(g, clip):
    n = torch.linalg.vector_norm(g)
    inv = torch.where(n > 0, 1.0 / n, torch.full_like(n, float("inf")))
    return g * (clip * torch.minimum(inv, torch.full_like(n, 1.0 / clip)))


class GanRnnTorchTwin:
    """Single-tower GAN_RNN on torch-CPU (gan_rnn_placeholder.py:191-298)."""

    def __init__(self, cfg: NetCfg, g_params: Dict[str, np.ndarray], d_params: Dict[str, np.ndarray], *,
                 g_learning_rate=8e-5, d_learning_rate=1e-3, mse_lambda=10.0, l2_scale=0.0,
                 clip_norm=15.0, dtype=torch.float32):
        self.cfg = cfg
        self.dtype = dtype
        self.g = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in g_params.items()}
        self.d = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in d_params.items()}
        self.g_learning_rate, self.d_learning_rate = g_learning_rate, d_learning_rate
        self.mse_lambda, self.l2_scale, self.clip_norm = mse_lambda, l2_scale, clip_norm
        self.d_real, self.d_fake = 1.0, 0.0
        self.beta1, self.beta2, self.eps = 0.9, 0.999, 1e-8
        self.m = {k: torch.zeros_like(v) for k, v in self.g.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.g.items()}
        self.t = 0

    def _t(self, a):
        return torch.as_tensor(np.asarray(a), dtype=self.dtype)

    def forward(self, inputs, lengths):
        with torch.no_grad():
            return generator(self.cfg, self.g, self._t(inputs), torch.as_tensor(np.asarray(lengths)).int()).numpy()

    def d_losses_and_grads(self, inputs, labels, lengths, noise_real=None, noise_fake=None):
        x, lab = self._t(inputs), self._t(labels)
        ln = torch.as_tensor(np.asarray(lengths)).int()
        with torch.no_grad():
            y = generator(self.cfg, self.g, x, ln)
        lr_ = discriminator(self.cfg, self.d, lab, ln, None if noise_real is None else self._t(noise_real))
        lf_ = discriminator(self.cfg, self.d, y, ln, None if noise_fake is None else self._t(noise_fake))
        d_rl = ((lr_ - self.d_real) ** 2).mean()
        d_fk = ((lf_ - self.d_fake) ** 2).mean()
        d_loss = d_rl + d_fk
        names = list(self.d)
        grads = torch.autograd.grad(d_loss, [self.d[k] for k in names])
        return (float(d_rl.detach()), float(d_fk.detach()), float(d_loss.detach())), dict(zip(names, grads))

    def g_losses_and_grads(self, inputs, labels, lengths, noise_fake=None):
        x, lab = self._t(inputs), self._t(labels)
        ln = torch.as_tensor(np.asarray(lengths)).int()
        y = generator(self.cfg, self.g, x, ln)
        lf_ = discriminator(self.cfg, self.d, y, ln, None if noise_fake is None else self._t(noise_fake))
        g_adv = ((lf_ - self.d_real) ** 2).mean()
        mse = 0.5 * self.cfg.output_dim * ((y - lab) ** 2).mean()
        if self.l2_scale > 0:
            l2 = self.l2_scale * sum(0.5 * (v ** 2).sum() for k, v in self.g.items() if "bias" not in k)
        else:
            l2 = torch.zeros((), dtype=self.dtype)
        g_loss = g_adv + self.mse_lambda * mse + l2
        names = list(self.g)
        grads = torch.autograd.grad(g_loss, [self.g[k] for k in names])
        return (float(g_adv.detach()), float(mse.detach()), float(l2.detach()), float(g_loss.detach())), dict(zip(names, grads)), y.detach().numpy()

    def d_step(self, inputs, labels, lengths, noise_real=None, noise_fake=None):
        losses, grads = self.d_losses_and_grads(inputs, labels, lengths, noise_real, noise_fake)
        with torch.no_grad():
            for k, v in self.d.items():
                v -= self.d_learning_rate * _clip_by_norm(grads[k], self.clip_norm)
        return losses

    def g_step(self, inputs, labels, lengths, noise_fake=None):
        losses, grads, _ = self.g_losses_and_grads(inputs, labels, lengths, noise_fake)
        self.t += 1
        lr_t = self.g_learning_rate * math.sqrt(1 - self.beta2 ** self.t) / (1 - self.beta1 ** self.t)
        with torch.no_grad():
            for k, p in self.g.items():
                g = _clip_by_norm(grads[k], self.clip_norm)
                self.m[k].mul_(self.beta1).add_(g, alpha=1 - self.beta1)
                self.v[k].mul_(self.beta2).addcmul_(g, g, value=1 - self.beta2)
                p -= lr_t * self.m[k] / (self.v[k].sqrt() + self.eps)
        return losses
