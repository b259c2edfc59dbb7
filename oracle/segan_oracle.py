"""CPU ORACLE (test infrastructure only) for the SEGAN-style conv G/D of the reference: models/segan.py:SEGAN with
models/generator.py:AEGenerator, models/discriminator.py:discriminator, utils/bnorm.py:VBN and the conv helpers of
utils/ops.py (downconv :78-100, deconv :277-311, conv1d :136-154, prelu :123-134, leakyrelu :120-121,
gaussian_noise_layer :19-30).  BASELINE.json configs[4].

*** TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__ and bench.py's cpu_baseline.
*** PARITY UNPINNED w.r.t. the reference: TensorFlow 1.4 cannot run here, and the shipped trainer cannot run anywhere
*** (models/segan.py:136 calls an undefined variables_on_gpu0(); scripts/train_segan.py:20 imports a missing utils.utils --
*** SURVEY 0-D7).  The graph itself is fully specified; this file restates it in torch float64 (autograd gives the gradients).
*** What pins it: direct-loop numpy convolutions for the TF SAME / conv2d_transpose index arithmetic, closed-form VBN and
*** RMSProp known answers, finite differences (tests/test_oracle_segan.py).

The graph (segan.py:155-236; one tower), with Lx = input_dim*(left+1+right) "samples" and U = output_dim:
  inputs [B, Lx], labels [B, U]
  G = AEGenerator(inputs[..., None])                                             generator.py:112-295, shipped flags run_segan.sh:96-124
      enc_i: h = downconv(h, depth_i, kwidth=20, stride 2, SAME) + b ; skip_i = h (i < n-1, BEFORE the activation) ; h = prelu(h)
      h = concat([z, h], channels), z ~ N(0,1) [B, len(h), depth_{n-1}] drawn per run (generator.py:201-205; injected here)
      dec_j: h = deconv(h, out_len = len(skip) or Lx, depth, kwidth=20, stride 2) + b ; j < n-1: h = concat([prelu(h), skip], channels)
      last : G = dense(h[..., 0], U)[..., None]   (tf.layers.dense over the sample axis: kernel [Lx, U], bias [U])
  D(joint [B, Lx+U]) = + noise -> 11 x [downconv k=31 + b -> VBN -> leakyrelu(0.3)] -> conv1d k=31, 1 kernel, no bias -> FC(len -> 1)
      called three times: the "dummy" pass on concat(inputs, labels) creates the VBN objects (reference batch = the fed batch, with
      its OWN noise draw; its mean / mean-of-squares are graph tensors, so gradients flow into it), then real = concat(inputs, labels)
      and fake = concat(inputs, G) mix their own batch statistics with the reference ones by 1/(B+1)         bnorm.py:36-48
  d_rl = mean((D(real)-1)^2), d_fk = mean(D(fake)^2), d_loss = d_rl + d_fk
  g_adv = mean((D(fake)-1)^2), g_l1 = l1_lambda * mean|G - labels|, g_loss = g_adv + g_l1                    segan.py:226-235
  both nets: tf.train.RMSPropOptimizer(lr) = decay 0.9, momentum 0, epsilon 1e-10, rms slot initialised to ONES   segan.py:123-124
      ms = 0.9 ms + 0.1 g^2 ; var -= lr * g / sqrt(ms + 1e-10)       (TF 1.4 ApplyRMSProp), no gradient clipping; towers averaged.
TF SAME padding for stride s, width k, length L: out = ceil(L/s), total = max((out-1)*s + k - L, 0), left = total // 2.
conv2d_transpose(x [Lin] -> [Lout], stride 2, SAME) is the gradient of that convolution: y[i] = sum x[o] W[dk] over 2o + dk - left = i.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

DEPTHS = (16, 32, 32, 64, 64, 128, 128, 256, 256, 512, 1024)       # segan.py:89,91: g_enc_depths = d_num_fmaps


@dataclass
class SeganCfg:
    input_len: int = 2827              # input_dim * (left_context + 1 + right_context) = 257 * 11 (run_segan.sh:99-102)
    output_dim: int = 40
    g_depths: Tuple[int, ...] = DEPTHS
    d_depths: Tuple[int, ...] = DEPTHS
    g_kwidth: int = 20                 # generator.py:151
    d_kwidth: int = 31                 # discriminator.py:79,88
    g_nl: str = "prelu"                # run_segan.sh:120
    lrelu_alpha: float = 0.3           # utils/ops.py:120
    vbn_eps: float = 1e-5              # bnorm.py:17


def same_pad(L, k, s=2):
    out = -(-L // s)
    total = max((out - 1) * s + k - L, 0)
    return out, total // 2, total - total // 2


def enc_lengths(L, n):
    out = [L]
    for _ in range(n):
        out.append(-(-out[-1] // 2))
    return out


def g_param_specs(cfg: SeganCfg) -> List[Tuple[str, Tuple[int, ...]]]:
    s, cin = [], 1
    n = len(cfg.g_depths)
    for i, d in enumerate(cfg.g_depths):
        s += [("g_ae/enc_%d/W" % i, (cfg.g_kwidth, 1, cin, d)), ("g_ae/enc_%d/b" % i, (d,))]
        if cfg.g_nl == "prelu":
            s.append(("g_ae/enc_prelu_%d/alpha" % i, (d,)))
        cin = d
    cin = 2 * cfg.g_depths[-1]                                  # concat([z, code])
    dec = list(cfg.g_depths[:-1][::-1]) + [1]
    for j, d in enumerate(dec):
        s += [("g_ae/dec_%d/W" % j, (cfg.g_kwidth, 1, d, cin)), ("g_ae/dec_%d/b" % j, (d,))]
        if j < n - 1:
            if cfg.g_nl == "prelu":
                s.append(("g_ae/dec_prelu_%d/alpha" % j, (d,)))
            cin = 2 * d                                         # concat([h, skip])
    s += [("g_ae/dense/kernel", (cfg.input_len, cfg.output_dim)), ("g_ae/dense/bias", (cfg.output_dim,))]
    return s


def d_param_specs(cfg: SeganCfg) -> List[Tuple[str, Tuple[int, ...]]]:
    s, cin = [], 1
    for i, d in enumerate(cfg.d_depths):
        s += [("d_model/d_block_%d/downconv/W" % i, (cfg.d_kwidth, 1, cin, d)), ("d_model/d_block_%d/downconv/b" % i, (d,)),
              ("d_model/d_block_%d/d_vbn_%d/gamma" % (i, i), (d,)), ("d_model/d_block_%d/d_vbn_%d/beta" % (i, i), (d,))]
        cin = d
    Ld = enc_lengths(cfg.input_len + cfg.output_dim, len(cfg.d_depths))[-1]
    s += [("d_model/logits_conv/W", (cfg.d_kwidth, cin, 1)), ("d_model/fully_connected/weights", (Ld, 1)),
          ("d_model/fully_connected/biases", (1,))]
    return s


def init_params(specs, rng, dtype=np.float64):
    """truncated_normal(0.02) conv filters, zero biases, alpha 0, gamma ~ N(1, 0.02), beta 0, xavier-uniform dense / FC
    (generator.py:170-171,286-287; discriminator.py:49-50,88-91; bnorm.py:53-57)."""
    out = {}
    for name, shape in specs:
        if name.endswith("/W"):
            w = rng.normal(0, 0.02, shape)
            out[name] = np.clip(w, -0.04, 0.04).astype(dtype)
        elif name.endswith("gamma"):
            out[name] = rng.normal(1.0, 0.02, shape).astype(dtype)
        elif name.endswith("kernel") or name.endswith("weights"):
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            out[name] = rng.uniform(-lim, lim, shape).astype(dtype)
        else:
            out[name] = np.zeros(shape, dtype)
    return out


# ------------------------------------------------------------------ layers (torch, [B, L, C] channels-last like the reference)
def downconv(x, W, b, stride=2):
    """utils/ops.py:78-100: tf.nn.conv2d on [B, L, 1, Cin] with W [k, 1, Cin, Cout], strides [1, 2, 1, 1], SAME."""
    k = W.shape[0]
    _, pl, pr = same_pad(x.shape[1], k, stride)
    xp = F.pad(x.permute(0, 2, 1), (pl, pr))
    y = F.conv1d(xp, W[:, 0].permute(2, 1, 0), stride=stride)       # torch weight [Cout, Cin, k]
    y = y.permute(0, 2, 1)
    return y + b if b is not None else y


def conv1d_same(x, W):
    """utils/ops.py:136-154: tf.nn.conv1d stride 1 SAME, W [k, Cin, Cout]."""
    k = W.shape[0]
    _, pl, pr = same_pad(x.shape[1], k, 1)
    xp = F.pad(x.permute(0, 2, 1), (pl, pr))
    return F.conv1d(xp, W.permute(2, 1, 0)).permute(0, 2, 1)


def deconv(x, W, b, out_len, stride=2):
    """utils/ops.py:277-311: tf.nn.conv2d_transpose with W [k, 1, Cout, Cin], output_shape [B, out_len, 1, Cout], SAME."""
    k = W.shape[0]
    lin, pl, _ = same_pad(out_len, k, stride)
    assert lin == x.shape[1], (lin, x.shape)
    full = F.conv_transpose1d(x.permute(0, 2, 1), W[:, 0].permute(2, 1, 0), stride=stride)     # torch weight [Cin, Cout, k]
    y = full[:, :, pl:pl + out_len].permute(0, 2, 1)
    return y + b if b is not None else y


def prelu(x, alpha):
    """utils/ops.py:123-134: relu(x) + alpha * (x - |x|) / 2."""
    return torch.relu(x) + alpha * (x - x.abs()) * 0.5


def leaky(x, a):
    return torch.maximum(x, a * x)


def vbn_stats(h):
    return h.mean(dim=(0, 1)), (h * h).mean(dim=(0, 1))


def vbn_apply(h, mean, mean_sq, gamma, beta, eps):
    return (h - mean) / torch.sqrt(eps + mean_sq - mean * mean) * gamma + beta


class SeganOracle:
    def __init__(self, cfg: SeganCfg, g: Dict[str, np.ndarray], d: Dict[str, np.ndarray], batch_size: int,
                 g_learning_rate=1e-3, d_learning_rate=1e-3, l1_lambda=100.0):
        self.cfg, self.B = cfg, batch_size
        self.g = {k: np.array(v, np.float64) for k, v in g.items()}
        self.d = {k: np.array(v, np.float64) for k, v in d.items()}
        self.g_lr, self.d_lr, self.l1_lambda = g_learning_rate, d_learning_rate, l1_lambda
        self.g_ms = {k: np.ones_like(v) for k, v in self.g.items()}        # RMSProp "rms" slot: ones (TF 1.4)
        self.d_ms = {k: np.ones_like(v) for k, v in self.d.items()}

    # ---- networks
    def _G(self, P, x, z):
        cfg = self.cfg
        n = len(cfg.g_depths)
        h = x[..., None]
        skips = []
        for i in range(n):
            h = downconv(h, P["g_ae/enc_%d/W" % i], P["g_ae/enc_%d/b" % i])
            if i < n - 1:
                skips.append(h)
            h = prelu(h, P["g_ae/enc_prelu_%d/alpha" % i]) if cfg.g_nl == "prelu" else leaky(h, cfg.lrelu_alpha)
        h = torch.cat([z, h], 2)
        for j in range(n):
            out_len = skips[-(j + 1)].shape[1] if j < n - 1 else x.shape[1]
            h = deconv(h, P["g_ae/dec_%d/W" % j], P["g_ae/dec_%d/b" % j], out_len)
            if j < n - 1:
                h = prelu(h, P["g_ae/dec_prelu_%d/alpha" % j]) if cfg.g_nl == "prelu" else leaky(h, cfg.lrelu_alpha)
                h = torch.cat([h, skips[-(j + 1)]], 2)
        return h[..., 0] @ P["g_ae/dense/kernel"] + P["g_ae/dense/bias"]

    def _D(self, P, joint, noise, ref=None):
        """one discriminator call; ref = None: the reference ("dummy") pass, returns (logits, [(mean, mean_sq)] per block)."""
        cfg = self.cfg
        h = (joint + noise)[..., None]
        stats = []
        for i in range(len(cfg.d_depths)):
            pre = "d_model/d_block_%d/" % i
            h = downconv(h, P[pre + "downconv/W"], P[pre + "downconv/b"])
            m, q = vbn_stats(h)
            if ref is not None:                                    # bnorm.py:36-48
                c = 1.0 / (self.B + 1.0)
                m, q = c * m + (1.0 - c) * ref[i][0], c * q + (1.0 - c) * ref[i][1]
            stats.append((m, q))
            h = vbn_apply(h, m, q, P[pre + "d_vbn_%d/gamma" % i], P[pre + "d_vbn_%d/beta" % i], cfg.vbn_eps)
            h = leaky(h, cfg.lrelu_alpha)
        h = conv1d_same(h, P["d_model/logits_conv/W"])[..., 0]
        return h @ P["d_model/fully_connected/weights"] + P["d_model/fully_connected/biases"], stats

    @staticmethod
    def _t(P, grad):
        return {k: torch.tensor(v, dtype=torch.float64, requires_grad=grad) for k, v in P.items()}

    def forward(self, x, z):
        with torch.no_grad():
            return self._G(self._t(self.g, False), torch.as_tensor(x, dtype=torch.float64), torch.as_tensor(z, dtype=torch.float64)).numpy()

    def _run(self, x, lab, z, n_ref, n_real, n_fake, which):
        tt = lambda a: torch.as_tensor(np.asarray(a, np.float64))
        x, lab, z = tt(x), tt(lab), tt(z)
        zero = torch.zeros(x.shape[0], x.shape[1] + lab.shape[1], dtype=torch.float64)
        n_ref, n_real, n_fake = [zero if n is None else tt(n) for n in (n_ref, n_real, n_fake)]
        G_, D_ = self._t(self.g, which == "g"), self._t(self.d, which == "d")
        Gx = self._G(G_, x, z)
        real, fake = torch.cat([x, lab], 1), torch.cat([x, Gx], 1)
        _, ref = self._D(D_, real, n_ref)                          # the dummy pass (segan.py:186-188)
        if which == "d":
            d_rl = ((self._D(D_, real, n_real, ref)[0] - 1.0) ** 2).mean()
            d_fk = (self._D(D_, fake, n_fake, ref)[0] ** 2).mean()
            loss = d_rl + d_fk
            out = [d_rl, d_fk, loss]
            P = D_
        else:
            g_adv = ((self._D(D_, fake, n_fake, ref)[0] - 1.0) ** 2).mean()
            g_l1 = self.l1_lambda * (Gx - lab).abs().mean()
            loss = g_adv + g_l1
            out = [g_adv, g_l1, loss]
            P = G_
        loss.backward()
        grads = {k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape)) for k, v in P.items()}
        return np.array([float(o.detach()) for o in out]), grads, Gx.detach().numpy()

    def d_tower(self, x, lab, z, n_ref=None, n_real=None, n_fake=None):
        l, g, _ = self._run(x, lab, z, n_ref, n_real, n_fake, "d")
        return l, g

    def g_tower(self, x, lab, z, n_ref=None, n_fake=None):
        return self._run(x, lab, z, n_ref, None, n_fake, "g")

    @staticmethod
    def _rmsprop(P, ms, grads, lr):
        for k in P:
            ms[k] = 0.9 * ms[k] + 0.1 * grads[k] ** 2
            P[k] = P[k] - lr * grads[k] / np.sqrt(ms[k] + 1e-10)

    def d_step(self, x, lab, z, n_ref=None, n_real=None, n_fake=None):
        l, g = self.d_tower(x, lab, z, n_ref, n_real, n_fake)
        self._rmsprop(self.d, self.d_ms, g, self.d_lr)
        return l

    def g_step(self, x, lab, z, n_ref=None, n_fake=None):
        l, g, _ = self.g_tower(x, lab, z, n_ref, n_fake)
        self._rmsprop(self.g, self.g_ms, g, self.g_lr)
        return l
