"""Pins for the frame-level DNN-GAN oracle: torch autograd + finite differences + closed forms."""
import math

import numpy as np
import torch

from oracle import dnn_gan_oracle as DO


def small():
    return DO.DnnCfg(input_dim=6, output_dim=4, left_context=1, right_context=1, g_units=10, g_hidden=3, d_units=9, d_hidden=2)


def params(cfg, seed=0):
    rng = np.random.default_rng(seed)
    g = DO.init_params(DO.g_param_specs(cfg), rng)
    d = DO.init_params(DO.d_param_specs(cfg), rng, relu_init=True)
    for p in (g, d):
        for k in p:
            if k.endswith("biases"):
                p[k] = rng.normal(0, 0.1, p[k].shape)
    return g, d


def test_specs_match_reference_sizes():
    cfg = DO.DnnCfg()
    gs, ds = DO.g_param_specs(cfg), DO.d_param_specs(cfg)
    assert [s for _, s in gs][0] == (2827, 1024) and gs[-2][1] == (1024, 40) and len(gs) == 10
    assert ds[0][1] == (297, 1024) and ds[-2][1] == (1024, 1) and len(ds) == 10
    assert gs[2][0] == "g_model/fully_connected_1/weights" and ds[-1][0] == "d_model/fully_connected_4/biases"


def _torch_losses(cfg, g, d, x, lab, lam, l2):
    G = {k: torch.tensor(v, requires_grad=True) for k, v in g.items()}
    D = {k: torch.tensor(v, requires_grad=True) for k, v in d.items()}
    X, L = torch.tensor(x), torch.tensor(lab)

    def stack(P, prefix, n, h):
        for i, name in enumerate(DO._fc_names(prefix, n)):
            h = h @ P[name + "/weights"] + P[name + "/biases"]
            if i < n - 1:
                h = torch.relu(h)
        return h
    y = stack(G, "g_model", cfg.g_hidden + 1, X)
    di = X[:, cfg.input_dim * cfg.left_context: cfg.input_dim * (cfg.left_context + 1)]
    dfun = lambda j: torch.clamp(stack(D, "d_model", cfg.d_hidden + 1, j), cfg.clip_lo, cfg.clip_hi)
    d_rl = ((dfun(torch.cat([di, L], 1)) - 1) ** 2).mean()
    d_fk = ((dfun(torch.cat([di, y.detach()], 1)) - 0) ** 2).mean()
    g_adv = ((dfun(torch.cat([di, y], 1)) - 1) ** 2).mean()
    g_mse = 0.5 * ((y - L) ** 2).mean() * cfg.output_dim
    g_l2 = l2 * sum(0.5 * (v ** 2).sum() for k, v in G.items() if k.endswith("weights"))
    return G, D, d_rl, d_fk, g_adv, g_mse, g_l2, g_adv + lam * g_mse + g_l2


def test_towers_match_autograd_including_clip_mask():
    cfg = small()
    g, d = params(cfg, 1)
    rng = np.random.default_rng(2)
    x = rng.normal(size=(7, cfg.fed_dim)); lab = rng.normal(size=(7, cfg.output_dim))
    for k in d:                                   # scale D so that some logits leave [-0.5, 1.5]
        if k.endswith("weights"):
            d[k] = d[k] * 2.0
    o = DO.GanDnnOracle(cfg, g, d, l2_scale=1e-3)
    G, D, d_rl, d_fk, g_adv, g_mse, g_l2, g_loss = _torch_losses(cfg, g, d, x, lab, 10.0, 1e-3)
    (rl, fk, dl), dg = o.d_tower(x, lab)
    assert np.allclose([rl, fk, dl], [d_rl.item(), d_fk.item(), (d_rl + d_fk).item()], rtol=1e-12)
    tg = torch.autograd.grad(d_rl + d_fk, list(D.values()))
    for k, t in zip(D, tg):
        assert np.allclose(dg[k], t.numpy(), rtol=1e-9, atol=1e-13), k
    (adv, mse, l2, gl), gg, _ = o.g_tower(x, lab)
    assert np.allclose([adv, mse, l2, gl], [g_adv.item(), g_mse.item(), g_l2.item(), g_loss.item()], rtol=1e-12)
    tg = torch.autograd.grad(g_loss, list(G.values()))
    for k, t in zip(G, tg):
        assert np.allclose(gg[k], t.numpy(), rtol=1e-9, atol=1e-13), k
    raw = DO.d_forward(cfg, o.d, np.concatenate([o._d_inputs(x), lab], 1))[1]
    assert (raw > cfg.clip_hi).any() or (raw < cfg.clip_lo).any()           # the clip branch was exercised


def test_adam_both_nets_no_clipping_and_first_step_sign():
    cfg = small()
    g, d = params(cfg, 3)
    rng = np.random.default_rng(4)
    x = rng.normal(size=(5, cfg.fed_dim)) * 30; lab = rng.normal(size=(5, cfg.output_dim)) * 30     # huge gradients
    o = DO.GanDnnOracle(cfg, g, d, g_learning_rate=1e-3, d_learning_rate=2e-3)
    _, dg = o.d_tower(x, lab)
    d0 = {k: v.copy() for k, v in o.d.items()}
    o.d_step(x, lab)
    for k in d0:
        big = np.abs(dg[k]) > 1e-2
        assert np.allclose((o.d[k] - d0[k])[big], -2e-3 * np.sign(dg[k][big]), rtol=1e-3), k    # Adam, not SGD; unclipped
    o.g_step(x, lab); o.g_step(x, lab)
    assert o.adam["g"]["t"] == 2 and o.adam["d"]["t"] == 1
    e = o.ema["d"]
    assert all(np.allclose(e[k], 0.9999 * d0[k] + 1e-4 * o.d[k]) for k in d0)


def test_dropout_towers_match_autograd_with_the_same_masks():
    """keep_prob < 1 (dnn.py:86,99,116-121; discriminator_dnn.py:68,81): y = relu(.) / keep * mask after every hidden layer of both
    nets, separate masks per run, per net, per layer and per D call; inert with l2_scale = 0 and in evaluation fetches
    (dnn.py:67-71).  Masks are an input of the oracle; torch autograd on the same masks pins losses and gradients."""
    cfg = small()
    g, d = params(cfg, 4)
    rng = np.random.default_rng(5)
    N, keep, lam, l2 = 9, 0.75, 10.0, 1e-3
    x = rng.normal(size=(N, cfg.fed_dim)); lab = rng.normal(size=(N, cfg.output_dim))
    masks = {}

    def mask_fn(run, net, layer, call, rows, cols):
        k = (run, net, layer, call)
        if k not in masks:
            masks[k] = (np.random.default_rng(hash(k) & 0xFFFF).random((rows, cols)) < keep).astype(np.float64)
        return masks[k]
    o = DO.GanDnnOracle(cfg, g, d, mse_lambda=lam, l2_scale=l2, keep_prob=keep, mask_fn=mask_fn)

    def tstack(P, prefix, n, h, run, net, call):
        for i, name in enumerate(DO._fc_names(prefix, n)):
            h = h @ P[name + "/weights"] + P[name + "/biases"]
            if i < n - 1:
                h = torch.relu(h) / keep * torch.tensor(masks[(run, net, i, call)])
        return h
    # D-run = training run 1: G forward, D(real) = call 0, D(fake) = call 1
    (d_rl, d_fk, d_loss), dg = o.d_tower(x, lab)
    assert sorted(masks) == [(1, 0, i, 0) for i in range(cfg.g_hidden)] + [(1, 1, i, c) for i in range(cfg.d_hidden) for c in (0, 1)]
    G = {k: torch.tensor(v, requires_grad=True) for k, v in g.items()}
    D = {k: torch.tensor(v, requires_grad=True) for k, v in d.items()}
    X, L = torch.tensor(x), torch.tensor(lab)
    di = X[:, cfg.input_dim * cfg.left_context: cfg.input_dim * (cfg.left_context + 1)]
    y = tstack(G, "g_model", cfg.g_hidden + 1, X, 1, 0, 0)
    dfun = lambda j, run, call: torch.clamp(tstack(D, "d_model", cfg.d_hidden + 1, j, run, 1, call), cfg.clip_lo, cfg.clip_hi)
    t_rl = ((dfun(torch.cat([di, L], 1), 1, 0) - 1) ** 2).mean()
    t_fk = (dfun(torch.cat([di, y.detach()], 1), 1, 1) ** 2).mean()
    assert math.isclose(d_rl, t_rl.item(), rel_tol=1e-12) and math.isclose(d_fk, t_fk.item(), rel_tol=1e-12)
    (t_rl + t_fk).backward()
    for k in dg:
        assert np.allclose(dg[k], D[k].grad.numpy(), rtol=1e-10, atol=1e-14), k
    # G-run = training run 2: new masks for G and for D(fake)
    (g_adv, g_mse, g_l2, g_loss), gg, _ = o.g_tower(x, lab)
    assert (2, 0, 0, 0) in masks and (2, 1, 0, 1) in masks and (2, 1, 0, 0) not in masks
    G = {k: torch.tensor(v, requires_grad=True) for k, v in g.items()}
    y = tstack(G, "g_model", cfg.g_hidden + 1, X, 2, 0, 0)
    t_adv = ((dfun(torch.cat([di, y], 1), 2, 1) - 1) ** 2).mean()
    t_mse = 0.5 * ((y - L) ** 2).mean() * cfg.output_dim
    t_l2 = l2 * sum(0.5 * (v ** 2).sum() for k, v in G.items() if k.endswith("weights"))
    assert math.isclose(g_adv, t_adv.item(), rel_tol=1e-12) and math.isclose(g_mse, t_mse.item(), rel_tol=1e-12)
    (t_adv + lam * t_mse + t_l2).backward()
    for k in gg:
        assert np.allclose(gg[k], G[k].grad.numpy(), rtol=1e-10, atol=1e-14), k
    # evaluation fetches and l2_scale = 0 do not drop
    n = len(masks)
    plain = DO.GanDnnOracle(cfg, g, d, mse_lambda=lam, l2_scale=l2)
    assert np.allclose(o.d_step(x, lab, train=False), plain.d_step(x, lab, train=False), rtol=1e-12)
    o0 = DO.GanDnnOracle(cfg, g, d, mse_lambda=lam, l2_scale=0.0, keep_prob=keep, mask_fn=mask_fn)
    p0 = DO.GanDnnOracle(cfg, g, d, mse_lambda=lam, l2_scale=0.0)
    assert np.allclose(o0.d_tower(x, lab)[0], p0.d_tower(x, lab)[0], rtol=1e-12)
    assert len(masks) == n
