"""CPU ORACLE (test infrastructure only) for the R-CED generator (models/rced.py:RCED) under DNNTrainer
(models/dnn_trainer.py) or paired with discriminator_dnn as in models/gan.py.

*** TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__ and bench.py's cpu_baseline.
*** PARITY UNPINNED w.r.t. the reference (TensorFlow 1.4 cannot run here); pinned by finite differences and by an
*** independent direct-loop convolution (tests/test_oracle_rced.py).

`RcedCfg.batch_norm=True` (run_dnn.sh:134): every conv2d becomes relu(batch_norm(conv(x))) without a bias, statistics per output
channel over [N, S, W] (oracle/bn_renorm.py); the linear output layer keeps its bias; is_training = not cross_validation
(rced.py:37,60-61).

Restated graph (batch_norm=False; rced.py:36-112):
  inputs [N, S*W] -> reshape [N, S, W, 1]   (S = left+1+right spliced frames = height, W = input_dim = width)
  9 x tf.contrib.layers.conv2d(num_outputs = 12,16,20,24,32,24,20,16,12; kernel [S, 13,11,9,7,7,7,9,11,13]; stride 1;
      padding SAME (the contrib default); ReLU; weights xavier, biases zero)        rced.py:90-102
  reshape [N, S*W*12] -> fully_connected(output_dim), linear, biases init 0.1          rced.py:110-116
SAME padding with an odd kernel k pads (k-1)//2 on both sides (S and every width are odd here; an even S pads the extra
row at the bottom, as TF does).  Variables (TF auto-numbered scopes): g_model/Conv{,_1..8}/weights [S, fw, Cin, Cout],
.../biases [Cout], g_model/fully_connected/weights [S*W*12, Dout], /biases [Dout].
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

from . import bn_renorm as bn
from .dnn_gan_oracle import DnnCfg, GanDnnOracle

FILTERS_NUM = (12, 16, 20, 24, 32, 24, 20, 16, 12)        # rced.py:92
FILTERS_WIDTH = (13, 11, 9, 7, 7, 7, 9, 11, 13)            # rced.py:93


@dataclass
class RcedCfg(DnnCfg):
    filters_num: Tuple[int, ...] = FILTERS_NUM
    filters_width: Tuple[int, ...] = FILTERS_WIDTH

    @property
    def splice(self):
        return self.left_context + 1 + self.right_context


def _conv_names(n):
    return ["g_model/Conv" + ("" if i == 0 else "_%d" % i) for i in range(n)]


def g_param_specs(cfg: RcedCfg) -> List[Tuple[str, Tuple[int, ...]]]:
    s, cin = [], 1
    for name, co, fw in zip(_conv_names(len(cfg.filters_num)), cfg.filters_num, cfg.filters_width):
        s.append((name + "/weights", (cfg.splice, fw, cin, co)))
        # contrib conv2d under normalizer_fn=batch_norm (rced.py:67-72,97-99): no biases, <scope>/BatchNorm/* over the channel axis
        s += bn.var_specs(name, co) if cfg.batch_norm else [(name + "/biases", (co,))]
        cin = co
    s += [("g_model/fully_connected/weights", (cfg.splice * cfg.input_dim * cin, cfg.output_dim)),
          ("g_model/fully_connected/biases", (cfg.output_dim,))]
    return s


def init_params(specs, rng, dtype=np.float64):
    """xavier_initializer() uniform: conv fan_in = kh*kw*Cin, fan_out = kh*kw*Cout; FC biases 0.1 (rced.py:116)."""
    out = {}
    for name, shape in specs:
        if "/BatchNorm/" in name:
            one = name.endswith("gamma") or name.endswith("moving_variance")
            out[name] = (np.ones if one else np.zeros)(shape, dtype)
        elif name.endswith("biases"):
            out[name] = np.full(shape, 0.1 if "fully_connected" in name else 0.0, dtype)
        elif len(shape) == 4:
            rf = shape[0] * shape[1]
            lim = math.sqrt(6.0 / (rf * shape[2] + rf * shape[3]))
            out[name] = rng.uniform(-lim, lim, shape).astype(dtype)
        else:
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            out[name] = rng.uniform(-lim, lim, shape).astype(dtype)
    return out


def _pads(k):
    return (k - 1) // 2, k - 1 - (k - 1) // 2


def im2col(x, kh, kw):
    """x [N, S, W, C] -> [N*S*W, kh*kw*C] with SAME zero padding, column order (dh, dw, c) = the filter's row-major order."""
    N, S, W, C = x.shape
    (pt, pb), (pl, pr) = _pads(kh), _pads(kw)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    win = np.lib.stride_tricks.sliding_window_view(xp, (kh, kw), axis=(1, 2))     # [N, S, W, C, kh, kw]
    return np.ascontiguousarray(win.transpose(0, 1, 2, 4, 5, 3)).reshape(N * S * W, kh * kw * C)


def col2im(dcol, shape, kh, kw):
    """Adjoint of im2col: scatter-add [N*S*W, kh*kw*C] back to [N, S, W, C]."""
    N, S, W, C = shape
    (pt, pb), (pl, pr) = _pads(kh), _pads(kw)
    dxp = np.zeros((N, S + pt + pb, W + pl + pr, C), dcol.dtype)
    d6 = dcol.reshape(N, S, W, kh, kw, C)
    for dh in range(kh):
        for dw in range(kw):
            dxp[:, dh:dh + S, dw:dw + W, :] += d6[:, :, :, dh, dw, :]
    return dxp[:, pt:pt + S, pl:pl + W, :]


def rced_fwd(cfg: RcedCfg, P, x, training=True):
    N = x.shape[0]
    S, W = cfg.splice, cfg.input_dim
    h = x.reshape(N, S, W, 1)
    cache = []
    for name, fw in zip(_conv_names(len(cfg.filters_num)), cfg.filters_width):
        Wt = P[name + "/weights"]
        col = im2col(h, S, fw)
        bcache = None
        if name + "/BatchNorm/beta" in P:          # moments over [N, S, W] per channel = over the rows of the [N*S*W, C] matrix
            z = col @ Wt.reshape(-1, Wt.shape[3])
            if training:
                z, bcache = bn.forward_train(P, name, z)
            else:
                z = bn.forward_infer(P, name, z)
        else:
            z = col @ Wt.reshape(-1, Wt.shape[3]) + P[name + "/biases"]
        a = np.maximum(z, 0.0)
        cache.append((h.shape, col, a, bcache))
        h = a.reshape(N, S, W, Wt.shape[3])
    flat = h.reshape(N, -1)
    y = flat @ P["g_model/fully_connected/weights"] + P["g_model/fully_connected/biases"]
    return y, (cache, flat)


def rced_bwd(cfg: RcedCfg, P, cache, dy, trace=None, start=None):
    """Gradients of every variable from dy = d loss / d y.  Test hooks: `trace` (a dict) receives, per conv layer, the gradient
    w.r.t. the layer's ReLU OUTPUT (before the mask); `start = (layer, d)` skips everything above `layer` and back-propagates
    the given gradient w.r.t. that layer's PRE-activation instead (the linear response to flipping ReLU masks there:
    tests/test_gpu_trainers.py::test_rced_reference_frame_random_relu_masks)."""
    convs, flat = cache
    S = cfg.splice
    names = _conv_names(len(cfg.filters_num))
    if start is None:
        grads = {"g_model/fully_connected/weights": flat.T @ dy, "g_model/fully_connected/biases": dy.sum(0)}
        d = (dy @ P["g_model/fully_connected/weights"].T).reshape(-1, cfg.filters_num[-1])
        top = len(names) - 1
    else:
        grads = {}
        top, d = start
    for i in range(top, -1, -1):
        in_shape, col, a, bcache = convs[i]
        Wt = P[names[i] + "/weights"]
        if start is None or i < top:
            if trace is not None:
                trace[i] = d.copy()
            d = d * (a > 0)
        if bcache is not None:
            d, gb = bn.backward_train(P, bcache, d)
            grads.update(gb)
        elif names[i] + "/biases" in P:
            grads[names[i] + "/biases"] = d.sum(0)
        grads[names[i] + "/weights"] = (col.T @ d).reshape(Wt.shape)
        if i > 0:
            dcol = d @ Wt.reshape(-1, Wt.shape[3]).T
            d = col2im(dcol, in_shape, S, cfg.filters_width[i]).reshape(-1, in_shape[3])
    return grads


class GanRcedOracle(GanDnnOracle):
    """GanDnnOracle with the R-CED generator: `supervised = True` gives the DNNTrainer graph (dnn_trainer.py:98-99,139-148)."""

    def _g_fwd(self, x):
        return rced_fwd(self.cfg, self.g, x, self.training)

    def _g_commit(self, cache, times):
        for _ in range(times):
            for conv in cache[0]:
                if conv[3] is not None:
                    bn.commit(self.g, conv[3])

    def _g_bwd(self, cache, dy):
        return rced_bwd(self.cfg, self.g, cache, dy)
