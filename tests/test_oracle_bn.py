"""Pins oracle/bn_renorm.py (the restatement of TF-1.4 batch_norm(renorm=True, scale=True)) with closed-form known answers,
torch autograd and finite differences, and the update schedule of oracle/dnn_gan_oracle.py with batch_norm=True."""
import numpy as np
import pytest
import torch

from oracle import bn_renorm as bn
from oracle.dnn_gan_oracle import DnnCfg, GanDnnOracle, d_param_specs, g_param_specs, init_params, trainable


def _layer(C, rng=None):
    P = bn.init_vars("L", C)
    if rng is not None:
        P["L/BatchNorm/gamma"] = rng.uniform(0.5, 1.5, C)
        P["L/BatchNorm/beta"] = rng.normal(0, 0.3, C)
    return P


def test_first_call_is_plain_batch_norm_and_first_commit_is_the_batch():
    rng = np.random.default_rng(0)
    x = rng.normal(2.0, 3.0, (50, 7))
    P = _layer(7)
    y, cache = bn.forward_train(P, "L", x)
    # zero-initialised renorm state: mixed moments are the batch's own, so r = 1, d = 0 exactly
    assert np.array_equal(cache["r"], np.ones(7)) and np.array_equal(cache["d"], np.zeros(7))
    mean, var = x.mean(0), x.var(0)
    assert np.allclose(y, (x - mean) / np.sqrt(var + 1e-3), rtol=0, atol=1e-12)
    bn.commit(P, cache)
    p = "L/BatchNorm/"
    assert np.allclose(P[p + "renorm_mean"], 0.01 * mean) and np.isclose(P[p + "renorm_mean_weight"], 0.01)
    assert np.allclose(P[p + "renorm_stddev"], 0.01 * np.sqrt(var + 1e-3)) and np.isclose(P[p + "renorm_stddev_weight"], 0.01)
    # debiased averages after one update are the batch's moments; moving_* start at (0, 1) with decay 0.999
    assert np.allclose(P[p + "moving_mean"], 0.001 * mean)
    assert np.allclose(P[p + "moving_variance"], 1.0 - 0.001 * (1.0 - var))


def test_second_call_corrections_closed_form():
    rng = np.random.default_rng(1)
    x1, x2 = rng.normal(0.5, 1.0, (40, 5)), rng.normal(-1.0, 2.0, (40, 5))
    P = _layer(5, rng)
    _, c1 = bn.forward_train(P, "L", x1)
    bn.commit(P, c1)
    y2, c2 = bn.forward_train(P, "L", x2)
    m1, s1 = x1.mean(0), np.sqrt(x1.var(0) + 1e-3)
    m2, s2 = x2.mean(0), np.sqrt(x2.var(0) + 1e-3)
    mixed_m, mixed_s = 0.01 * m1 + 0.99 * m2, 0.01 * s1 + 0.99 * s2
    assert np.allclose(c2["r"], s2 / mixed_s) and np.allclose(c2["d"], (m2 - mixed_m) / mixed_s)
    g, b = P["L/BatchNorm/gamma"], P["L/BatchNorm/beta"]
    assert np.allclose(y2, ((x2 - m2) / s2 * c2["r"] + c2["d"]) * g + b)
    # forward_train leaves the state alone until commit
    assert np.isclose(P["L/BatchNorm/renorm_mean_weight"], 0.01)
    bn.commit(P, c2)
    assert np.isclose(P["L/BatchNorm/renorm_mean_weight"], 0.01 + 0.99 * 0.01)
    new_mean = (0.99 * 0.01 * m1 + 0.01 * m2) / (0.0199)
    assert np.allclose(P["L/BatchNorm/moving_mean"], 0.999 * (0.001 * m1) + 0.001 * new_mean)


def test_inference_uses_moving_statistics():
    rng = np.random.default_rng(2)
    P = _layer(4, rng)
    P["L/BatchNorm/moving_mean"] = rng.normal(0, 1, 4)
    P["L/BatchNorm/moving_variance"] = rng.uniform(0.5, 2, 4)
    x = rng.normal(0, 1, (9, 4))
    y = bn.forward_infer(P, "L", x)
    ref = (x - P["L/BatchNorm/moving_mean"]) / np.sqrt(P["L/BatchNorm/moving_variance"] + 1e-3) * P["L/BatchNorm/gamma"] + P["L/BatchNorm/beta"]
    assert np.allclose(y, ref)


def test_backward_matches_torch_autograd_with_stopped_corrections():
    rng = np.random.default_rng(3)
    C, n = 6, 33
    P = _layer(C, rng)
    p = "L/BatchNorm/"
    P[p + "renorm_mean"] = rng.normal(0, 0.02, C); P[p + "renorm_mean_weight"] = np.float64(0.3)
    P[p + "renorm_stddev"] = rng.uniform(0.2, 0.4, C); P[p + "renorm_stddev_weight"] = np.float64(0.3)
    x = rng.normal(0.3, 1.5, (n, C))
    dy = rng.normal(0, 1, (n, C))
    y, cache = bn.forward_train(P, "L", x)
    dx, grads = bn.backward_train(P, cache, dy)
    xt = torch.tensor(x, requires_grad=True)
    gt = torch.tensor(P[p + "gamma"], requires_grad=True)
    bt = torch.tensor(P[p + "beta"], requires_grad=True)
    mean = xt.mean(0)
    var = ((xt - mean) ** 2).mean(0)
    std = torch.sqrt(var + 1e-3)
    mixed_m = torch.tensor(P[p + "renorm_mean"]) + (1 - 0.3) * mean
    mixed_s = torch.tensor(P[p + "renorm_stddev"]) + (1 - 0.3) * std
    r, d = (std / mixed_s).detach(), ((mean - mixed_m) / mixed_s).detach()
    yt = (xt - mean) / std * (r * gt) + (d * gt + bt)
    assert np.allclose(yt.detach().numpy(), y, atol=1e-12)
    yt.backward(torch.tensor(dy))
    assert np.allclose(xt.grad.numpy(), dx, atol=1e-10)
    assert np.allclose(gt.grad.numpy(), grads[p + "gamma"], atol=1e-10)
    assert np.allclose(bt.grad.numpy(), grads[p + "beta"], atol=1e-10)


def _tiny(batch_norm=True, seed=5):
    cfg = DnnCfg(input_dim=6, output_dim=4, left_context=1, right_context=1, g_units=10, g_hidden=2, d_units=9, d_hidden=2,
                 batch_norm=batch_norm)
    rng = np.random.default_rng(seed)
    g = init_params(g_param_specs(cfg), rng)
    d = init_params(d_param_specs(cfg), rng, relu_init=True)
    x = rng.normal(0, 1, (12, cfg.fed_dim))
    lab = rng.normal(0, 1, (12, cfg.output_dim))
    return cfg, g, d, x, lab


def test_specs_with_batch_norm():
    cfg, g, d, _, _ = _tiny()
    names = [n for n, _ in g_param_specs(cfg)]
    assert names[:9] == ["g_model/fully_connected/weights"] + ["g_model/fully_connected/BatchNorm/" + k for k in
                                                                ("beta", "gamma", "moving_mean", "moving_variance", "renorm_mean",
                                                                 "renorm_mean_weight", "renorm_stddev", "renorm_stddev_weight")]
    assert names[-2:] == ["g_model/fully_connected_2/weights", "g_model/fully_connected_2/biases"]     # the linear output keeps its bias
    assert not any(n.endswith("biases") for n in names[:-2])
    assert sum(trainable(n) for n in names) == 2 * 3 + 2


def test_gradients_at_initial_state_match_finite_differences():
    """With zero renorm state r = 1 and d = 0 for ANY batch, so the stop_gradient'ed loss equals the plain loss there."""
    cfg, g, d, x, lab = _tiny()
    o = GanDnnOracle(cfg, g, d, mse_lambda=3.0)
    (_, _, d_loss), dgrads = GanDnnOracle(cfg, g, d, mse_lambda=3.0).d_tower(x, lab)
    (_, _, _, g_loss), ggrads, _ = GanDnnOracle(cfg, g, d, mse_lambda=3.0).g_tower(x, lab)
    rng = np.random.default_rng(9)
    h = 1e-6
    for which, grads, params, idx in (("d", dgrads, d, 2), ("g", ggrads, g, 3)):
        for name in [n for n in params if trainable(n)]:
            k = tuple(rng.integers(0, s) for s in params[name].shape)
            vals = []
            for sgn in (+1, -1):
                pp = {a: b.copy() for a, b in params.items()}
                pp[name][k] += sgn * h
                oo = GanDnnOracle(cfg, pp if which == "g" else g, pp if which == "d" else d, mse_lambda=3.0)
                vals.append(oo.d_tower(x, lab, want_grads=False)[0][2] if which == "d" else oo.g_tower(x, lab, want_grads=False)[0][3])
            fd = (vals[0] - vals[1]) / (2 * h)
            assert abs(fd - grads[name][k]) <= 1e-6 + 1e-5 * abs(fd), (name, fd, grads[name][k])


def test_update_schedule_of_one_run():
    cfg, g, d, x, lab = _tiny()
    o = GanDnnOracle(cfg, g, d)
    o.d_step(x, lab)
    # generator: two identical updates; discriminator: real, real, fake
    w2 = 1.0 - 0.99 ** 2
    assert np.isclose(o.g["g_model/fully_connected/BatchNorm/renorm_mean_weight"], w2)
    assert np.isclose(o.d["d_model/fully_connected/BatchNorm/renorm_stddev_weight"], 1.0 - 0.99 ** 3)
    z = x @ g["g_model/fully_connected/weights"]
    assert np.allclose(o.g["g_model/fully_connected/BatchNorm/renorm_mean"], w2 * z.mean(0))
    # the state is not a trainable variable: Adam never touches it, and its slots stay zero
    assert np.all(o.adam["g"]["m"]["g_model/fully_connected/BatchNorm/moving_mean"] == 0)
    before = {k: v.copy() for k, v in o.g.items()}
    o.g_step(x, lab)
    assert np.isclose(o.g["g_model/fully_connected/BatchNorm/renorm_mean_weight"], 1.0 - 0.99 ** 4)
    assert np.isclose(o.d["d_model/fully_connected/BatchNorm/renorm_stddev_weight"], 1.0 - 0.99 ** 6)
    assert not np.allclose(before["g_model/fully_connected/BatchNorm/gamma"], o.g["g_model/fully_connected/BatchNorm/gamma"])


def test_cross_validation_twin_uses_moving_statistics_and_leaves_state():
    cfg, g, d, x, lab = _tiny()
    o = GanDnnOracle(cfg, g, d)
    for _ in range(3):
        o.d_step(x, lab); o.g_step(x, lab)
    cv = GanDnnOracle(cfg, o.g, o.d, cross_validation=True)
    snap = {k: v.copy() for k, v in cv.g.items()}
    l1 = cv.g_step(x, lab, train=False)
    l2 = cv.g_step(x[:5], lab[:5], train=False)
    assert all(np.array_equal(snap[k], cv.g[k]) for k in snap)
    # inference statistics: a row's output does not depend on the rest of the batch
    assert np.allclose(cv.forward(x)[:5], cv.forward(x[:5]))
    assert np.isfinite(l1).all() and np.isfinite(l2).all()


def test_rced_with_batch_norm_specs_gradients_and_schedule():
    from oracle import rced_oracle as R
    cfg = R.RcedCfg(input_dim=7, output_dim=3, left_context=1, right_context=1, filters_num=(3, 4, 2), filters_width=(5, 3, 3), batch_norm=True)
    specs = R.g_param_specs(cfg)
    names = [n for n, _ in specs]
    assert names[:3] == ["g_model/Conv/weights", "g_model/Conv/BatchNorm/beta", "g_model/Conv/BatchNorm/gamma"]
    assert names[-2:] == ["g_model/fully_connected/weights", "g_model/fully_connected/biases"] and len(names) == 3 * 9 + 2
    assert dict(specs)["g_model/Conv_1/BatchNorm/renorm_stddev"] == (4,) and dict(specs)["g_model/Conv_1/BatchNorm/renorm_stddev_weight"] == ()
    rng = np.random.default_rng(4)
    g = R.init_params(specs, rng)
    for k in g:
        if k.endswith("/beta"):
            g[k] = rng.normal(0, 0.2, g[k].shape)
        elif k.endswith("/gamma"):
            g[k] = rng.uniform(0.6, 1.4, g[k].shape)
    x = rng.standard_normal((5, cfg.fed_dim)); lab = rng.standard_normal((5, cfg.output_dim))

    def make(params):
        o = R.GanRcedOracle(cfg, params, {}, mse_lambda=1.0, l2_scale=1e-2)
        o.supervised = True
        return o
    losses, grads, y = make(g).g_tower(x, lab)
    # zero renorm state: r = 1, d = 0 for any batch, so central differences of the loss see the same function
    for name in [n for n in g if trainable(n)]:
        idx = tuple(rng.integers(0, s) for s in g[name].shape)
        vals = []
        for sgn in (1, -1):
            pp = {k: v.copy() for k, v in g.items()}
            pp[name][idx] += sgn * 1e-6
            vals.append(make(pp).g_tower(x, lab, want_grads=False)[0][3])
        fd = (vals[0] - vals[1]) / 2e-6
        assert abs(fd - grads[name][idx]) < 1e-6 * max(1.0, abs(fd)) + 1e-8, (name, fd, grads[name][idx])
    o = make(g)
    o.g_step(x, lab)
    assert np.isclose(o.g["g_model/Conv_2/BatchNorm/renorm_mean_weight"], 1 - 0.99 ** 2)       # the generator's call twice
    # per-channel moments over [N, S, W]
    col = R.im2col(x.reshape(5, 3, 7, 1), 3, 5)
    z = col @ g["g_model/Conv/weights"].reshape(-1, 3)
    assert np.allclose(o.g["g_model/Conv/BatchNorm/renorm_mean"], (1 - 0.99 ** 2) * z.mean(0))
