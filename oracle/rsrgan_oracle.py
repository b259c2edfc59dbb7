"""CPU ORACLE (test infrastructure only) for RSRGAN's sequence-level GAN step.

*** THIS FILE IS TEST INFRASTRUCTURE.  It is never imported by the product
*** path (`rsrgan_amd/`); only `tests/`, `__graft_entry__.smoke()` and
*** `bench.py`'s `cpu_baseline` leg may import it, and there only as the
*** checker.

*** PARITY UNPINNED: the reference (Python 2.7 + TensorFlow 1.4) cannot run in
*** this image and ships no tests, golden vectors or fixtures for this path
*** (SURVEY.md section 8c).  This restatement therefore follows the reference
*** source line by line plus TF-1.4's documented semantics, and is pinned only
*** by (a) analytic known-answer tests, (b) torch.nn.LSTM(proj_size=...) on
*** the peephole-free subset, (c) torch-autograd and finite-difference checks
*** of the hand-written backward pass (tests/test_oracle_*.py).

Plain NumPy; every function cites the reference file:line it restates.  All
arrays are batch-major [B, T, D] exactly like the reference placeholders
(models/gan_rnn_placeholder.py:94-104).  `dtype` is float64 for the parity
truth and float32 for like-for-like checks.

The arithmetic itself lives in the un-vendored dependency TensorFlow 1.4.0
(README.md:16).  TF semantics restated here:
  * tf.contrib.rnn.LSTMCell(use_peepholes, num_proj, forget_bias=1.0): gate
    order i, j, f, o; forget_bias added at run time; projection has no bias
    (in-repo statement of the same cell: models/BNLSTMCell.py:160-217).
  * tf.nn.dynamic_rnn(sequence_length): for t >= len[b] the output is zero and
    the (c, m) state is copied through.
  * contrib.layers.fully_connected contracts the last axis, weights [in,out].
  * tf.clip_by_norm is per tensor: t * clip * min(1/||t||, 1/clip).
  * tf.train.AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); eps outside sqrt.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


# ----------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------
@dataclass
class NetCfg:
    """Shapes of G and D.  Defaults = what the reference hard-codes.

    g_type 'lstm'          : models/lstm.py:43-45   (FC 257->280, 3xLSTMP(760,p280), FC->40)
    g_type 'res_lstm_l'    : models/res_lstm_l.py:43-45 (4xLSTMP(760,p257) + running residual)
    g_type 'res_lstm_base' : same stack, no residual adds
    D                      : models/discriminator_lstm.py:26-28 (2xLSTMP(256,p40), FC->1)
    A projection of 0 means num_proj=None (m = h), used by BASELINE.json's
    "2x512" naming.
    """
    input_dim: int = 257
    output_dim: int = 40
    g_type: str = "lstm"
    g_layers: int = 3
    g_cells: int = 760
    g_proj: int = 280
    d_type: str = "lstm"              # "dnn" = models/discriminator_dnn.py applied per frame to the 40-dim target
    d_layers: int = 2
    d_cells: int = 256
    d_proj: int = 40
    forget_bias: float = 1.0
    lrelu_alpha: float = 0.3          # utils/ops.py:120-121

    @staticmethod
    def res_lstm_l(**kw) -> "NetCfg":
        c = NetCfg(g_type="res_lstm_l", g_layers=4, g_cells=760, g_proj=257)
        for k, v in kw.items():
            setattr(c, k, v)
        return c


def _rec_dim(cells: int, proj: int) -> int:
    return proj if proj > 0 else cells


def _lstmp_specs(prefix: str, in_dim: int, cells: int, proj: int) -> List[Tuple[str, Tuple[int, ...]]]:
    """Variables of one tf.contrib.rnn.LSTMCell in TF-1.4 creation order."""
    r = _rec_dim(cells, proj)
    s = [(prefix + "/kernel", (in_dim + r, 4 * cells)),
         (prefix + "/bias", (4 * cells,)),
         (prefix + "/w_f_diag", (cells,)),
         (prefix + "/w_i_diag", (cells,)),
         (prefix + "/w_o_diag", (cells,))]
    if proj > 0:
        s.append((prefix + "/projection/kernel", (cells, proj)))
    return s


def g_param_specs(cfg: NetCfg) -> List[Tuple[str, Tuple[int, ...]]]:
    """g_vars in graph-construction order (models/lstm.py:82-124,
    models/res_lstm_l.py:101-194)."""
    s: List[Tuple[str, Tuple[int, ...]]] = []
    r = _rec_dim(cfg.g_cells, cfg.g_proj)
    if cfg.g_type == "lstm":
        s += [("g_model/fully_connected/weights", (cfg.input_dim, r)),
              ("g_model/fully_connected/biases", (r,))]
        for l in range(cfg.g_layers):
            s += _lstmp_specs("g_model/rnn/multi_rnn_cell/cell_%d/lstm_cell" % l,
                              r, cfg.g_cells, cfg.g_proj)
        s += [("g_model/fully_connected_1/weights", (r, cfg.output_dim)),
              ("g_model/fully_connected_1/biases", (cfg.output_dim,))]
    elif cfg.g_type in ("res_lstm_l", "res_lstm_base"):
        if cfg.g_type == "res_lstm_l":
            assert r == cfg.input_dim, "residual adds need proj == input_dim (res_lstm_l.py:111)"
        in_dim = cfg.input_dim
        for l in range(cfg.g_layers):
            s += _lstmp_specs("g_model/lstm_cell_%d/rnn/lstm_cell" % (l + 1),
                              in_dim, cfg.g_cells, cfg.g_proj)
            in_dim = r
        s += [("g_model/forward_out/fully_connected/weights", (r, cfg.output_dim)),
              ("g_model/forward_out/fully_connected/biases", (cfg.output_dim,))]
    else:
        raise ValueError("Unrecognized G type {}".format(cfg.g_type))  # gan_rnn_placeholder.py:131-132
    return s


def d_param_specs(cfg: NetCfg) -> List[Tuple[str, Tuple[int, ...]]]:
    """d_vars (models/discriminator_lstm.py:70-104 | models/discriminator_dnn.py:61-92)."""
    s: List[Tuple[str, Tuple[int, ...]]] = []
    if cfg.d_type == "dnn":
        dims = [cfg.output_dim] + [cfg.d_cells] * cfg.d_layers + [1]
        for i in range(cfg.d_layers + 1):
            n = "d_model/fully_connected" + ("" if i == 0 else "_%d" % i)
            s += [(n + "/weights", (dims[i], dims[i + 1])), (n + "/biases", (dims[i + 1],))]
        return s
    in_dim = cfg.output_dim
    r = _rec_dim(cfg.d_cells, cfg.d_proj)
    for l in range(cfg.d_layers):
        s += _lstmp_specs("d_model/rnn/multi_rnn_cell/cell_%d/lstm_cell" % l,
                          in_dim, cfg.d_cells, cfg.d_proj)
        in_dim = r
    s += [("d_model/fully_connected/weights", (r, 1)),
          ("d_model/fully_connected/biases", (1,))]
    return s


def xavier_init(specs, rng: np.random.Generator, dtype=np.float64) -> Dict[str, np.ndarray]:
    """xavier_initializer() (uniform) for weights/kernels/peepholes, zeros for
    biases (models/lstm.py:86-87,93; TF LSTMCell bias is zero-initialised)."""
    out: Dict[str, np.ndarray] = {}
    for name, shape in specs:
        if name.endswith("bias") or name.endswith("biases"):
            out[name] = np.zeros(shape, dtype)
        else:
            fan_in, fan_out = (shape[0], shape[0]) if len(shape) == 1 else (shape[0], shape[1])
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            out[name] = rng.uniform(-lim, lim, size=shape).astype(dtype)
    return out


# ----------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------
def sigmoid(x):
    return 0.5 * (np.tanh(0.5 * x) + 1.0)


def leakyrelu(x, alpha=0.3):
    """utils/ops.py:120-121: tf.maximum(x, alpha*x)."""
    return np.maximum(x, alpha * x)


def fc_fwd(x, w, b):
    """contrib.layers.fully_connected on rank-3 input (models/lstm.py:82-87)."""
    return x @ w + b


def fc_bwd(x, w, dy):
    x2 = x.reshape(-1, x.shape[-1])
    dy2 = dy.reshape(-1, dy.shape[-1])
    return dy @ w.T, x2.T @ dy2, dy2.sum(0)


def lstmp_fwd(x, lengths, p, forget_bias=1.0):
    """One dynamic_rnn(LSTMCell(use_peepholes, num_proj)) over [B,T,I].

    Cell math: models/BNLSTMCell.py:176-217 with the three batch_norm calls
    removed and W_xh;W_hh stacked into `kernel`.  Masking: TF dynamic_rnn.
    p = (kernel, bias, w_f, w_i, w_o, proj_or_None).
    Returns (out [B,T,R], cache)."""
    K, b, wf, wi, wo, Wp = p
    B, T, I = x.shape
    H = wf.shape[0]
    R = Wp.shape[1] if Wp is not None else H
    dt = x.dtype
    c = np.zeros((B, H), dt)
    m = np.zeros((B, R), dt)
    out = np.zeros((B, T, R), dt)
    steps = []
    for t in range(T):
        mask = (t < lengths)[:, None]
        xm = np.concatenate([x[:, t], m], axis=1)
        z = xm @ K + b
        i, j, f, o = z[:, :H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:]
        gi = sigmoid(i + wi * c)
        gf = sigmoid(f + forget_bias + wf * c)
        gj = np.tanh(j)
        cn = gf * c + gi * gj
        go = sigmoid(o + wo * cn)
        tc = np.tanh(cn)
        h = go * tc
        mn = h @ Wp if Wp is not None else h
        out[:, t] = np.where(mask, mn, 0.0)
        steps.append((mask, xm, c, gi, gj, gf, go, cn, tc, h))
        c = np.where(mask, cn, c)
        m = np.where(mask, mn, m)
    return out, (steps, I, H, R)


def lstmp_bwd(dout, cache, p):
    """BPTT of lstmp_fwd.  Returns (dx [B,T,I], grads tuple matching p)."""
    K, b, wf, wi, wo, Wp = p
    steps, I, H, R = cache
    B, T = dout.shape[0], dout.shape[1]
    dt = dout.dtype
    dK = np.zeros_like(K); db = np.zeros_like(b)
    dwf = np.zeros_like(wf); dwi = np.zeros_like(wi); dwo = np.zeros_like(wo)
    dWp = np.zeros_like(Wp) if Wp is not None else None
    dx = np.zeros((B, T, I), dt)
    dc = np.zeros((B, H), dt)
    dm = np.zeros((B, R), dt)
    for t in range(T - 1, -1, -1):
        mask, xm, cp, gi, gj, gf, go, cn, tc, h = steps[t]
        dm_tot = np.where(mask, dout[:, t] + dm, 0.0)
        dm_pass = np.where(mask, 0.0, dm)
        dc_new = np.where(mask, dc, 0.0)
        dc_pass = np.where(mask, 0.0, dc)
        if Wp is not None:
            dh = dm_tot @ Wp.T
            dWp += h.T @ dm_tot
        else:
            dh = dm_tot
        dao = dh * tc * go * (1.0 - go)
        dcn = dc_new + dh * go * (1.0 - tc * tc) + dao * wo
        dwo += (dao * cn).sum(0)
        daf = dcn * cp * gf * (1.0 - gf)
        dai = dcn * gj * gi * (1.0 - gi)
        dj = dcn * gi * (1.0 - gj * gj)
        dwf += (daf * cp).sum(0)
        dwi += (dai * cp).sum(0)
        dcp = dcn * gf + dai * wi + daf * wf
        dz = np.concatenate([dai, dj, daf, dao], axis=1)
        dK += xm.T @ dz
        db += dz.sum(0)
        dxm = dz @ K.T
        dx[:, t] = dxm[:, :I]
        dm = dxm[:, I:] + dm_pass
        dc = dcp + dc_pass
    return dx, (dK, db, dwf, dwi, dwo, dWp)


# ----------------------------------------------------------------------------
# networks
# ----------------------------------------------------------------------------
def _layer_params(params, prefix, has_proj):
    return (params[prefix + "/kernel"], params[prefix + "/bias"],
            params[prefix + "/w_f_diag"], params[prefix + "/w_i_diag"],
            params[prefix + "/w_o_diag"],
            params[prefix + "/projection/kernel"] if has_proj else None)


def _put_layer_grads(grads, prefix, g):
    names = ["/kernel", "/bias", "/w_f_diag", "/w_i_diag", "/w_o_diag", "/projection/kernel"]
    for n, v in zip(names, g):
        if v is not None:
            grads[prefix + n] = v


def generator_fwd(cfg: NetCfg, params, x, lengths, drop=None):
    """LSTM.infer (models/lstm.py:41-129) / RES_LSTM_L.infer
    (models/res_lstm_l.py:41-199).  x [B,T,Din] -> y [B,T,Dout].
    drop = (keep_prob, mask_of_layer(l) -> {0,1} array [B,T,P]) or None: tf.contrib.rnn.DropoutWrapper(cell,
    output_keep_prob) around every layer (lstm.py:99-102, res_lstm_l.py:96-99) -- the layer's OUTPUT sequence times mask / keep
    (nn_ops.dropout: div(x, keep) * mask; a fresh mask per time step), the carried state untouched; finished rows output zeros
    either way.  TF's random stream cannot be reproduced: the masks are an input."""
    hp = cfg.g_proj > 0
    cache = {}

    def dropped(l, out):
        if drop is None:
            return out
        keep, mask_of = drop
        m = np.asarray(mask_of(l), out.dtype)
        cache["drop%d" % l] = (m, keep)
        return out / keep * m
    if cfg.g_type == "lstm":
        a = fc_fwd(x, params["g_model/fully_connected/weights"], params["g_model/fully_connected/biases"])
        h = leakyrelu(a, cfg.lrelu_alpha)
        cache["x"], cache["a"] = x, a
        ins = [h]
        for l in range(cfg.g_layers):
            pre = "g_model/rnn/multi_rnn_cell/cell_%d/lstm_cell" % l
            out, c = lstmp_fwd(ins[-1], lengths, _layer_params(params, pre, hp), cfg.forget_bias)
            cache[pre] = c
            ins.append(dropped(l, out))
        cache["ins"] = ins
        y = fc_fwd(ins[-1], params["g_model/fully_connected_1/weights"], params["g_model/fully_connected_1/biases"])
    else:
        res = cfg.g_type == "res_lstm_l"
        ins = [x]
        for l in range(cfg.g_layers):
            pre = "g_model/lstm_cell_%d/rnn/lstm_cell" % (l + 1)
            out, c = lstmp_fwd(ins[-1], lengths, _layer_params(params, pre, hp), cfg.forget_bias)
            cache[pre] = c
            out = dropped(l, out)
            ins.append(out + ins[-1] if res else out)     # res_lstm_l.py:111,121,131,190
        cache["ins"] = ins
        y = fc_fwd(ins[-1], params["g_model/forward_out/fully_connected/weights"],
                   params["g_model/forward_out/fully_connected/biases"])
    return y, cache


def generator_bwd(cfg: NetCfg, params, cache, dy):
    hp = cfg.g_proj > 0
    grads = {}
    ins = cache["ins"]

    def through_drop(l, d):                      # gradient of div(x, keep) * mask
        dr = cache.get("drop%d" % l)
        return d if dr is None else d * dr[0] / dr[1]
    if cfg.g_type == "lstm":
        d, dw, db = fc_bwd(ins[-1], params["g_model/fully_connected_1/weights"], dy)
        grads["g_model/fully_connected_1/weights"], grads["g_model/fully_connected_1/biases"] = dw, db
        for l in range(cfg.g_layers - 1, -1, -1):
            pre = "g_model/rnn/multi_rnn_cell/cell_%d/lstm_cell" % l
            d, g = lstmp_bwd(through_drop(l, d), cache[pre], _layer_params(params, pre, hp))
            _put_layer_grads(grads, pre, g)
        a = cache["a"]
        da = d * np.where(a > 0, 1.0, cfg.lrelu_alpha)
        _, dw, db = fc_bwd(cache["x"], params["g_model/fully_connected/weights"], da)
        grads["g_model/fully_connected/weights"], grads["g_model/fully_connected/biases"] = dw, db
    else:
        res = cfg.g_type == "res_lstm_l"
        d, dw, db = fc_bwd(ins[-1], params["g_model/forward_out/fully_connected/weights"], dy)
        grads["g_model/forward_out/fully_connected/weights"] = dw
        grads["g_model/forward_out/fully_connected/biases"] = db
        for l in range(cfg.g_layers - 1, -1, -1):
            pre = "g_model/lstm_cell_%d/rnn/lstm_cell" % (l + 1)
            dx, g = lstmp_bwd(through_drop(l, d), cache[pre], _layer_params(params, pre, hp))
            _put_layer_grads(grads, pre, g)
            d = dx + d if res else dx
    return grads


def gaussian_noise_layer(x, noise):
    """utils/ops.py:19-30: x + N(0,std^2) with the noise tensor of shape
    [B,1,D] (static time dim None -> 1) broadcast over T.  The draw is an
    injected tensor here (None == std 0)."""
    return x if noise is None else x + noise


D_CLIP = (-0.5, 1.5)          # models/discriminator_dnn.py:93


def _dnn_names(cfg):
    return ["d_model/fully_connected" + ("" if i == 0 else "_%d" % i) for i in range(cfg.d_layers + 1)]


def discriminator_fwd(cfg: NetCfg, params, x, lengths, noise=None):
    """discriminator_lstm (models/discriminator_lstm.py:24-110), or discriminator_dnn
    (models/discriminator_dnn.py:21-98: no noise layer, ReLU FC stack, clip_by_value(-0.5, 1.5))."""
    if cfg.d_type == "dnn":
        acts = [x]
        names = _dnn_names(cfg)
        for i, n in enumerate(names):
            z = acts[-1] @ params[n + "/weights"] + params[n + "/biases"]
            acts.append(np.maximum(z, 0.0) if i < len(names) - 1 else z)
        return np.clip(acts[-1], *D_CLIP), {"acts": acts}
    hp = cfg.d_proj > 0
    cache = {}
    ins = [gaussian_noise_layer(x, noise)]
    for l in range(cfg.d_layers):
        pre = "d_model/rnn/multi_rnn_cell/cell_%d/lstm_cell" % l
        out, c = lstmp_fwd(ins[-1], lengths, _layer_params(params, pre, hp), cfg.forget_bias)
        cache[pre] = c
        ins.append(out)
    cache["ins"] = ins
    logits = fc_fwd(ins[-1], params["d_model/fully_connected/weights"], params["d_model/fully_connected/biases"])
    return logits, cache


def discriminator_bwd(cfg: NetCfg, params, cache, dlogits, want_param_grads=True):
    if cfg.d_type == "dnn":
        acts, names, grads = cache["acts"], _dnn_names(cfg), {}
        raw = acts[-1]
        d = dlogits * ((raw >= D_CLIP[0]) & (raw <= D_CLIP[1]))
        for i in range(len(names) - 1, -1, -1):
            if i < len(names) - 1:
                d = d * (acts[i + 1] > 0)
            a2 = acts[i].reshape(-1, acts[i].shape[-1]); d2 = d.reshape(-1, d.shape[-1])
            grads[names[i] + "/weights"] = a2.T @ d2
            grads[names[i] + "/biases"] = d2.sum(0)
            d = d @ params[names[i] + "/weights"].T
        return d, (grads if want_param_grads else None)
    hp = cfg.d_proj > 0
    grads = {}
    ins = cache["ins"]
    d, dw, db = fc_bwd(ins[-1], params["d_model/fully_connected/weights"], dlogits)
    grads["d_model/fully_connected/weights"], grads["d_model/fully_connected/biases"] = dw, db
    for l in range(cfg.d_layers - 1, -1, -1):
        pre = "d_model/rnn/multi_rnn_cell/cell_%d/lstm_cell" % l
        d, g = lstmp_bwd(d, cache[pre], _layer_params(params, pre, hp))
        _put_layer_grads(grads, pre, g)
    return d, (grads if want_param_grads else None)


# ----------------------------------------------------------------------------
# losses / optimizers  (models/gan_rnn_placeholder.py:244-260,144-147,178-184)
# ----------------------------------------------------------------------------
def lsgan_mean_sq(logits, target):
    """tf.reduce_mean(tf.squared_difference(logits, target)) over ALL B*T*1
    entries, padded frames included (gan_rnn_placeholder.py:244-246)."""
    d = logits - target
    return float(np.mean(d * d)), 2.0 * d / d.size


def g_mse(y, labels, output_dim):
    """0.5 * tf.losses.mean_squared_error(g, labels) * output_dim
    (gan_rnn_placeholder.py:252); mean over B*T*Dout incl. padded frames."""
    d = y - labels
    return float(0.5 * output_dim * np.mean(d * d)), output_dim * d / d.size


def l2_term(params, l2_scale):
    """gan_rnn_placeholder.py:253-258: sum of tf.nn.l2_loss over g_vars whose
    name does not contain 'bias'."""
    loss = 0.0
    grads = {}
    for n, v in params.items():
        if "bias" not in n:
            loss += 0.5 * float(np.sum(v * v))
            grads[n] = l2_scale * v
    return l2_scale * loss, grads


def clip_by_norm(g, clip):
    """tf.clip_by_norm(g, clip) per tensor (gan_rnn_placeholder.py:178-182)."""
    n = math.sqrt(float(np.sum(g.astype(np.float64) ** 2)))
    inv = (1.0 / n) if n > 0 else float("inf")
    return g * (clip * min(inv, 1.0 / clip))


def average_gradients(tower_grads: Sequence[Dict[str, np.ndarray]]):
    """utils/ops.py:343-376: per-variable mean over towers."""
    return {k: np.mean(np.stack([g[k] for g in tower_grads], 0), 0) for k in tower_grads[0]}


def exponential_decay(iteration, num_jobs, num_iters, init_learning_rate, multiply_jobs=True):
    """utils/ops.py:378-391."""
    final = 0.0001 * init_learning_rate
    if iteration + 1 >= num_iters:
        cur = final
    else:
        cur = init_learning_rate * math.exp(iteration * math.log(final / init_learning_rate) / num_iters)
    return num_jobs * cur if multiply_jobs else cur


# ----------------------------------------------------------------------------
# the model object: same surface train_one_iteration touches
# ----------------------------------------------------------------------------
class GanRnnOracle:
    """GAN_RNN (models/gan_rnn_placeholder.py:62-298) on one or more towers.

    `num_towers` reproduces the in-graph data parallelism (:139-189): the fed
    batch is sliced per tower, gradients are averaged per variable, clipped per
    tensor, then applied once (D: SGD :144, G: Adam :147)."""

    def __init__(self, cfg: NetCfg, g_params, d_params, *, batch_size, num_towers=1,
                 g_learning_rate=8e-5, d_learning_rate=1e-3, mse_lambda=10.0,
                 l2_scale=0.0, clip_norm=15.0, d_real=1.0, d_fake=0.0,
                 cross_validation=False, dtype=np.float64, keep_prob=1.0, mask_fn=None):
        self.cfg = cfg
        self.dtype = dtype
        # DropoutWrapper on the generator's layers: mask_fn(run, tower, layer, B, T, P) -> {0,1} [B,T,P]; run = index of the
        # training sess.run (1, 2, ...: every run draws new masks), is_training only (lstm.py:71-72)
        self.keep_prob, self.mask_fn, self._run = keep_prob, mask_fn, 0
        self.g = {k: np.array(v, dtype) for k, v in g_params.items()}
        self.d = {k: np.array(v, dtype) for k, v in d_params.items()}
        self.batch_size = batch_size
        self.num_towers = num_towers
        self.g_learning_rate = g_learning_rate
        self.d_learning_rate = d_learning_rate
        self.mse_lambda = mse_lambda
        self.l2_scale = l2_scale
        self.clip_norm = clip_norm
        self.d_real, self.d_fake = d_real, d_fake
        self.cross_validation = cross_validation
        self.beta1, self.beta2, self.eps = 0.9, 0.999, 1e-8
        self.adam_m = {k: np.zeros_like(v) for k, v in self.g.items()}
        self.adam_v = {k: np.zeros_like(v) for k, v in self.g.items()}
        self.adam_t = 0
        self.ema_decay = 0.9999
        self.g_ema = {k: v.copy() for k, v in self.g.items()}
        self.d_ema = {k: v.copy() for k, v in self.d.items()}

    # -- helpers -------------------------------------------------------------
    def _slice(self, a, k):
        return None if a is None else a[self.batch_size * k:self.batch_size * (k + 1)]

    def _drop(self, shape, training=True, tower=0):
        if not (training and self.keep_prob < 1.0 and not self.cross_validation):
            return None
        run, (B, T) = self._run, shape[:2]
        P = self.cfg.g_proj if self.cfg.g_proj > 0 else self.cfg.g_cells
        return self.keep_prob, (lambda l: self.mask_fn(run, tower, l, B, T, P))

    def forward(self, inputs, lengths):
        """model.g_outputs (gan_rnn_placeholder.py:133-135)."""
        x = np.asarray(inputs, self.dtype)
        y, _ = generator_fwd(self.cfg, self.g, x, np.asarray(lengths).astype(np.int32), self._drop(x.shape))
        return y

    # -- per-tower graphs (build_model_single_gpu :191-298) ---------------------
    def d_tower(self, x, lab, ln, noise_real=None, noise_fake=None, want_grads=True, tower=0):
        cfg = self.cfg
        y, _ = generator_fwd(cfg, self.g, x, ln, self._drop(x.shape, want_grads, tower))
        lr_, cr = discriminator_fwd(cfg, self.d, lab, ln, noise_real)
        lf_, cf = discriminator_fwd(cfg, self.d, y, ln, noise_fake)
        d_rl, dlr = lsgan_mean_sq(lr_, self.d_real)
        d_fk, dlf = lsgan_mean_sq(lf_, self.d_fake)
        grads = None
        if want_grads:
            _, gr = discriminator_bwd(cfg, self.d, cr, dlr)
            _, gf = discriminator_bwd(cfg, self.d, cf, dlf)
            grads = {k: gr[k] + gf[k] for k in gr}
        return (d_rl, d_fk, d_rl + d_fk), grads

    def g_tower(self, x, lab, ln, noise_fake=None, want_grads=True, tower=0):
        cfg = self.cfg
        y, cg = generator_fwd(cfg, self.g, x, ln, self._drop(x.shape, want_grads, tower))
        mse, dy_mse = g_mse(y, lab, cfg.output_dim)
        if getattr(self, "supervised", False):
            # models/rnn_trainer.py:146-156 (RNNTrainer): g_loss = g_mse + g_l2, no discriminator in the graph
            if (not self.cross_validation) and self.l2_scale > 0.0:
                g_l2, l2g = l2_term(self.g, self.l2_scale)
            else:
                g_l2, l2g = 0.0, {}
            grads = None
            if want_grads:
                grads = generator_bwd(cfg, self.g, cg, self.mse_lambda * dy_mse)
                for k, v in l2g.items():
                    grads[k] = grads[k] + v
            return (0.0, mse, g_l2, self.mse_lambda * mse + g_l2), grads, y
        lf_, cf = discriminator_fwd(cfg, self.d, y, ln, noise_fake)
        g_adv, dlf = lsgan_mean_sq(lf_, self.d_real)
        if (not self.cross_validation) and self.l2_scale > 0.0:
            g_l2, l2g = l2_term(self.g, self.l2_scale)
        else:
            g_l2, l2g = 0.0, {}
        g_loss = g_adv + self.mse_lambda * mse + g_l2
        grads = None
        if want_grads:
            dy_adv, _ = discriminator_bwd(cfg, self.d, cf, dlf, want_param_grads=False)
            grads = generator_bwd(cfg, self.g, cg, dy_adv + self.mse_lambda * dy_mse)
            for k, v in l2g.items():
                grads[k] = grads[k] + v
        return (g_adv, mse, g_l2, g_loss), grads, y

    # -- clip_by_norm per tensor (:178-182) then apply_gradients (:183-184), EMA (:185-186) --
    def apply_d(self, avg):
        for k in self.d:                                          # SGD :144,183
            self.d[k] = self.d[k] - self.d_learning_rate * clip_by_norm(avg[k], self.clip_norm)
            self.d_ema[k] = self.ema_decay * self.d_ema[k] + (1 - self.ema_decay) * self.d[k]

    def apply_g(self, avg):
        self.adam_t += 1                                          # Adam :147,184
        t = self.adam_t
        lr_t = self.g_learning_rate * math.sqrt(1 - self.beta2 ** t) / (1 - self.beta1 ** t)
        for k in self.g:
            g = clip_by_norm(avg[k], self.clip_norm)
            self.adam_m[k] = self.beta1 * self.adam_m[k] + (1 - self.beta1) * g
            self.adam_v[k] = self.beta2 * self.adam_v[k] + (1 - self.beta2) * g * g
            self.g[k] = self.g[k] - lr_t * self.adam_m[k] / (np.sqrt(self.adam_v[k]) + self.eps)
            self.g_ema[k] = self.ema_decay * self.g_ema[k] + (1 - self.ema_decay) * self.g[k]

    # -- sess.run([model.d_opt, ...]) (train_gan_rnn_placeholder.py:77-82) -------
    def d_step(self, inputs, labels, lengths, noise_real=None, noise_fake=None, train=True):
        x = np.asarray(inputs, self.dtype); lab = np.asarray(labels, self.dtype)
        ln = np.asarray(lengths).astype(np.int32)
        losses, tower_grads = [], []
        self._run += 1 if train else 0
        for k in range(self.num_towers):
            ls, g = self.d_tower(self._slice(x, k), self._slice(lab, k), self._slice(ln, k),
                                 self._slice(noise_real, k), self._slice(noise_fake, k), want_grads=train, tower=k)
            losses.append(ls); tower_grads.append(g)
        if train:
            self.apply_d(average_gradients(tower_grads))
        rl, fk, dl = zip(*losses)
        return list(rl), list(fk), list(dl)

    # -- sess.run([model.g_opt, ...]) (train_gan_rnn_placeholder.py:94-101) ------
    def g_step(self, inputs, labels, lengths, noise_fake=None, train=True):
        x = np.asarray(inputs, self.dtype); lab = np.asarray(labels, self.dtype)
        ln = np.asarray(lengths).astype(np.int32)
        losses, tower_grads = [], []
        self._run += 1 if train else 0
        for k in range(self.num_towers):
            ls, g, _ = self.g_tower(self._slice(x, k), self._slice(lab, k), self._slice(ln, k),
                                    self._slice(noise_fake, k), want_grads=train, tower=k)
            losses.append(ls); tower_grads.append(g)
        if train:
            self.apply_g(average_gradients(tower_grads))
        adv, mse, l2, gl = zip(*losses)
        return list(adv), list(mse), list(l2), list(gl)


def train_one_iteration(model: GanRnnOracle, batches, disc_updates=1, gen_updates=1, noises=None):
    """scripts/train_gan_rnn_placeholder.py:48-133 on a list of
    (inputs, labels, lengths) batches; returns the same 7 averages."""
    acc = np.zeros(7)
    d_counter = g_counter = 0
    model.d_real, model.d_fake = 1.0, 0.0                    # :63-64
    full = model.batch_size * model.num_towers
    for bi, (x, lab, ln) in enumerate(batches):
        if x.shape[0] != full:                               # :69-70
            continue
        nz = noises[bi] if noises is not None else {}
        for s in range(disc_updates):
            rl, fk, dl = model.d_step(x, lab, ln, nz.get(("d_real", s)), nz.get(("d_fake", s)))
            acc[0] += np.mean(rl); acc[1] += np.mean(fk); acc[2] += np.mean(dl)
            d_counter += 1
        for s in range(gen_updates):
            adv, mse, l2, gl = model.g_step(x, lab, ln, nz.get(("g_fake", s)))
            acc[3] += np.mean(adv); acc[4] += np.mean(mse); acc[5] += np.mean(l2); acc[6] += np.mean(gl)
            g_counter += 1
    acc[:3] /= max(d_counter, 1)
    acc[3:] /= max(g_counter, 1)
    return tuple(float(v) for v in acc)
