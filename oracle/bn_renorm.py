"""CPU ORACLE (test infrastructure only): `tf.contrib.layers.batch_norm(is_training, scale=True, renorm=True)` as the
reference's generators / discriminator call it (models/dnn.py:56-61, models/discriminator_dnn.py:36-41, models/rced.py:67-72).

*** TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__ and bench.py's cpu_baseline.
*** PARITY UNPINNED: the arithmetic lives in TensorFlow 1.4.0 (README.md:16), which is not under /root/reference and cannot be
*** installed here.  This file restates its published algorithm:
***   contrib/layers/python/layers/layers.py `batch_norm`: renorm=True is not fusable, `updates_collections=UPDATE_OPS`,
***   `batch_weights=None`, `zero_debias_moving_mean=False`  ->  the core layer `tf.layers.BatchNormalization(axis=-1,
***   momentum=decay=0.999, epsilon=0.001, center=True, scale=True, renorm=True, renorm_clipping=None,
***   renorm_momentum=renorm_decay=0.99)`;
***   python/layers/normalization.py `BatchNormalization.call` / `_renorm_correction_and_moments`.
*** and is pinned by closed-form known answers, by torch autograd and by finite differences (tests/test_oracle_bn.py).

Variables of one layer, in creation order (normalization.py `build`): beta [C] = 0, gamma [C] = 1 (trainable);
moving_mean [C] = 0, moving_variance [C] = 1, renorm_mean [C] = 0, renorm_mean_weight [] = 0, renorm_stddev [C] = 0,
renorm_stddev_weight [] = 0 (not trainable).

Training call on x [n, C] (statistics over every axis but the last):
    mean, var = moments(x)                       (biased variance)
    stddev    = sqrt(var + eps)
    mixed_mean   = renorm_mean   + (1 - renorm_mean_weight)   * mean
    mixed_stddev = renorm_stddev + (1 - renorm_stddev_weight) * stddev
    r = stop_gradient(stddev / mixed_stddev) ;  d = stop_gradient((mean - mixed_mean) / mixed_stddev)      (no clipping)
    y = (x - mean) / stddev * (r * gamma) + (d * gamma + beta)
  and, as UPDATE_OPS (assign_moving_average(v, value, decay, zero_debias=False): v -= (v - value) * (1 - decay)):
    renorm_mean  <- ema(renorm_mean, mean, 0.99),     renorm_mean_weight   <- ema(w, 1, 0.99),  new_mean   = renorm_mean / weight
    renorm_stddev<- ema(renorm_stddev, stddev, 0.99), renorm_stddev_weight <- ema(w, 1, 0.99),  new_stddev = renorm_stddev / weight
    moving_mean     <- ema(moving_mean, new_mean, 0.999)
    moving_variance <- ema(moving_variance, new_stddev^2 - eps, 0.999)
Inference call (is_training=False, the cross_validation twin): y = (x - moving_mean) / sqrt(moving_variance + eps) * gamma + beta.

Order of the updates inside one `sess.run`.  The reference collects EVERY update op of the graph under the gradients of both
optimizers (models/gan.py:139-146, models/dnn_trainer.py:124-128), and tower 0 builds the generator twice and the discriminator
three times (gan.py:162-181: the `reuse=False` pass, the dummy D, then the real ones), so one D-run or G-run executes, in an order
TensorFlow does not define: 2 x the generator's update (same batch), 2 x the discriminator's update with the real batch's moments
and 1 x with the fake batch's.  This restatement fixes one admissible serialisation: every correction (r, d) of a run reads the
state as it was when the run started, and `commit()` then applies the updates in the order the calls were made."""
from __future__ import annotations

import numpy as np

EPS = 1e-3
DECAY = 0.999
RENORM_DECAY = 0.99

STATE_NAMES = ("moving_mean", "moving_variance", "renorm_mean", "renorm_mean_weight", "renorm_stddev", "renorm_stddev_weight")
TRAINABLE_NAMES = ("beta", "gamma")


def var_specs(scope, C):
    """(name, shape) of the layer's variables under `<scope>/BatchNorm/`, creation order."""
    p = scope + "/BatchNorm/"
    return [(p + "beta", (C,)), (p + "gamma", (C,)), (p + "moving_mean", (C,)), (p + "moving_variance", (C,)),
            (p + "renorm_mean", (C,)), (p + "renorm_mean_weight", ()), (p + "renorm_stddev", (C,)), (p + "renorm_stddev_weight", ())]


def init_vars(scope, C, dtype=np.float64):
    out = {}
    for name, shape in var_specs(scope, C):
        one = name.endswith("gamma") or name.endswith("moving_variance")
        out[name] = (np.ones if one else np.zeros)(shape, dtype)
    return out


def is_state(name):
    return name.rsplit("/", 1)[-1] in STATE_NAMES


def _ema(v, value, decay):
    return v - (v - value) * (1.0 - decay)


def forward_train(P, scope, x):
    """x [n, C] (conv callers flatten [N, S, W, C] to [N*S*W, C]).  Returns (y, cache); the state in P is NOT touched
    (see commit)."""
    p = scope + "/BatchNorm/"
    mean = x.mean(0)
    var = ((x - mean) ** 2).mean(0)
    std = np.sqrt(var + EPS)
    mixed_mean = P[p + "renorm_mean"] + (1.0 - P[p + "renorm_mean_weight"]) * mean
    mixed_std = P[p + "renorm_stddev"] + (1.0 - P[p + "renorm_stddev_weight"]) * std
    r = std / mixed_std
    d = (mean - mixed_mean) / mixed_std
    xhat = (x - mean) / std
    y = xhat * (r * P[p + "gamma"]) + (d * P[p + "gamma"] + P[p + "beta"])
    return y, dict(scope=scope, xhat=xhat, std=std, mean=mean, r=r, d=d)


def backward_train(P, cache, dy):
    """Returns (dx, {beta, gamma grads}).  r and d are constants (stop_gradient); mean and var are differentiated."""
    p = cache["scope"] + "/BatchNorm/"
    xhat, std, r, d = cache["xhat"], cache["std"], cache["r"], cache["d"]
    n = dy.shape[0]
    s1 = dy.sum(0)
    s2 = (dy * xhat).sum(0)
    grads = {p + "beta": s1, p + "gamma": r * s2 + d * s1}
    dx = (P[p + "gamma"] * r / std) * (dy - s1 / n - xhat * (s2 / n))
    return dx, grads


def commit(P, cache):
    """The UPDATE_OPS of one call, applied to the state in P (in place)."""
    p = cache["scope"] + "/BatchNorm/"
    P[p + "renorm_mean"] = _ema(P[p + "renorm_mean"], cache["mean"], RENORM_DECAY)
    P[p + "renorm_mean_weight"] = _ema(P[p + "renorm_mean_weight"], 1.0, RENORM_DECAY)
    P[p + "renorm_stddev"] = _ema(P[p + "renorm_stddev"], cache["std"], RENORM_DECAY)
    P[p + "renorm_stddev_weight"] = _ema(P[p + "renorm_stddev_weight"], 1.0, RENORM_DECAY)
    new_mean = P[p + "renorm_mean"] / P[p + "renorm_mean_weight"]
    new_std = P[p + "renorm_stddev"] / P[p + "renorm_stddev_weight"]
    P[p + "moving_mean"] = _ema(P[p + "moving_mean"], new_mean, DECAY)
    P[p + "moving_variance"] = _ema(P[p + "moving_variance"], new_std * new_std - EPS, DECAY)


def forward_infer(P, scope, x):
    p = scope + "/BatchNorm/"
    return (x - P[p + "moving_mean"]) / np.sqrt(P[p + "moving_variance"] + EPS) * P[p + "gamma"] + P[p + "beta"]
