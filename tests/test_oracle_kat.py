"""Pins for the CPU oracle (it is 'parity unpinned' w.r.t. the reference, which
cannot run here): analytic known answers, torch.nn.LSTM(proj_size) on the
peephole-free subset, autograd twin and finite differences."""
import math

import numpy as np
import pytest
import torch

from oracle import rsrgan_oracle as O
from oracle import torch_twin as TT


def small_cfg(g_type="lstm", **kw):
    c = O.NetCfg(input_dim=9, output_dim=5, g_type=g_type, g_layers=2, g_cells=12, g_proj=7,
                 d_layers=2, d_cells=8, d_proj=5)
    if g_type != "lstm":
        c.g_proj = 9 if g_type == "res_lstm_l" else 7
        c.g_layers = 3
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def rand_params(cfg, seed=0, scale_bias=True):
    rng = np.random.default_rng(seed)
    g = O.xavier_init(O.g_param_specs(cfg), rng)
    d = O.xavier_init(O.d_param_specs(cfg), rng)
    if scale_bias:  # non-zero biases exercise more of the graph
        for p in (g, d):
            for k in p:
                if "bias" in k:
                    p[k] = rng.normal(0, 0.1, p[k].shape)
    return g, d


def test_param_counts_match_survey():
    # G-lstm 22 tensors, layer 1 920 520 params; D-lstm 14 tensors, 187 945 params
    cfg = O.NetCfg()
    gs = O.g_param_specs(cfg)
    assert len(gs) == 22
    layer = sum(int(np.prod(s)) for n, s in gs if "cell_0" in n)
    assert layer == 1920520
    ds = O.d_param_specs(cfg)
    assert len(ds) == 14
    assert sum(int(np.prod(s)) for _, s in ds) == 187945
    rs = O.g_param_specs(O.NetCfg.res_lstm_l())
    assert len(rs) == 26
    assert sum(int(np.prod(s)) for _, s in rs) == 7063120


def test_zero_weights_give_bias():
    cfg = small_cfg()
    g, d = rand_params(cfg)
    for k in g:
        if "bias" not in k:
            g[k][...] = 0
    x = np.random.default_rng(1).normal(size=(3, 6, cfg.input_dim))
    y, _ = O.generator_fwd(cfg, g, x, np.array([6, 4, 2]))
    assert np.allclose(y, g["g_model/fully_connected_1/biases"])


def test_masking_zero_output_and_frozen_state():
    cfg = small_cfg()
    g, _ = rand_params(cfg)
    rng = np.random.default_rng(2)
    x = rng.normal(size=(2, 8, 7))
    p = O._layer_params(g, "g_model/rnn/multi_rnn_cell/cell_0/lstm_cell", True)
    lens = np.array([8, 3])
    out, cache = O.lstmp_fwd(x, lens, p)
    assert np.all(out[1, 3:] == 0) and np.any(out[1, :3] != 0)
    # prefix of the short row equals the run truncated to 3 frames
    out3, _ = O.lstmp_fwd(x[:, :3], np.array([3, 3]), p)
    assert np.allclose(out[1, :3], out3[1])
    # carried state after t>=len equals state at len-1: c_prev at step 5 == cn at step 2
    steps = cache[0]
    assert np.allclose(steps[5][2][1], steps[2][7][1])


def test_lsgan_and_mse_closed_form():
    l = np.full((4, 5, 1), 0.25)
    v, dl = O.lsgan_mean_sq(l, 1.0)
    assert math.isclose(v, 0.5625) and np.allclose(dl, 2 * (-0.75) / 20)
    y = np.ones((2, 3, 40)); lab = np.zeros((2, 3, 40))
    v, dy = O.g_mse(y, lab, 40)
    assert math.isclose(v, 20.0) and np.allclose(dy, 1.0 / 6)   # d/dy = (y-lab)/(B*T)


def test_clip_by_norm():
    g = np.zeros(9); g[0] = 30.0
    assert np.allclose(O.clip_by_norm(g, 15.0), g / 2)
    g[0] = 3.0
    assert np.allclose(O.clip_by_norm(g, 15.0), g)
    assert np.all(O.clip_by_norm(np.zeros(4), 15.0) == 0)


def test_exponential_decay():
    assert math.isclose(O.exponential_decay(0, 1, 10, 8e-5), 8e-5)
    assert math.isclose(O.exponential_decay(9, 1, 10, 8e-5), 8e-9)
    assert math.isclose(O.exponential_decay(3, 2, 10, 1e-3), 2 * 1e-3 * math.exp(3 * math.log(1e-4) / 10))
    assert math.isclose(O.exponential_decay(3, 2, 10, 0.05, multiply_jobs=False), 0.05 * math.exp(3 * math.log(1e-4) / 10))


def test_adam_first_step_is_sign():
    cfg = small_cfg()
    g, d = rand_params(cfg)
    m = O.GanRnnOracle(cfg, g, d, batch_size=2, g_learning_rate=1e-3)
    rng = np.random.default_rng(3)
    x = rng.normal(size=(2, 5, 9)); lab = rng.normal(size=(2, 5, 5)); ln = np.array([5, 4])
    _, grads, _ = m.g_tower(x, lab, ln)
    before = {k: v.copy() for k, v in m.g.items()}
    m.g_step(x, lab, ln)
    for k in before:
        gk = grads[k]
        big = np.abs(gk) > 1e-3   # eps/sqrt(1-b2)=3.2e-7 perturbs smaller grads
        assert np.allclose((m.g[k] - before[k])[big], -1e-3 * np.sign(gk[big]), rtol=2e-3)


def test_lstmp_matches_torch_nn_lstm_without_peepholes():
    """Independent cross-check of gate order / forget bias / projection against
    torch.nn.LSTM(proj_size): TF order (i,j,f,o) -> torch (i,f,g,o)."""
    rng = np.random.default_rng(4)
    B, T, I, H, P = 3, 7, 6, 10, 4
    K = rng.normal(0, 0.3, (I + P, 4 * H)); b = rng.normal(0, 0.1, 4 * H); Wp = rng.normal(0, 0.3, (H, P))
    z = np.zeros(H)
    x = rng.normal(size=(B, T, I))
    out, _ = O.lstmp_fwd(x, np.full(B, T), (K, b, z, z, z, Wp), forget_bias=1.0)
    lstm = torch.nn.LSTM(I, H, proj_size=P, batch_first=True).double()
    perm = np.concatenate([np.arange(0, H), np.arange(2 * H, 3 * H), np.arange(H, 2 * H), np.arange(3 * H, 4 * H)])
    bb = b.copy(); bb[2 * H:3 * H] += 1.0
    with torch.no_grad():
        lstm.weight_ih_l0.copy_(torch.tensor(K[:I].T[perm]))
        lstm.weight_hh_l0.copy_(torch.tensor(K[I:].T[perm]))
        lstm.bias_ih_l0.copy_(torch.tensor(bb[perm]))
        lstm.bias_hh_l0.zero_()
        lstm.weight_hr_l0.copy_(torch.tensor(Wp.T))
        ref, _ = lstm(torch.tensor(x))
    assert np.allclose(out, ref.numpy(), atol=1e-12)


@pytest.mark.parametrize("g_type", ["lstm", "res_lstm_l", "res_lstm_base"])
def test_hand_bptt_matches_autograd(g_type):
    cfg = small_cfg(g_type)
    g, d = rand_params(cfg, seed=5)
    rng = np.random.default_rng(6)
    B, T = 4, 7
    x = rng.normal(size=(B, T, cfg.input_dim)); lab = rng.normal(size=(B, T, cfg.output_dim))
    ln = np.array([7, 5, 3, 1])
    nr = rng.normal(0, 0.05, (B, 1, cfg.output_dim)); nf = rng.normal(0, 0.05, (B, 1, cfg.output_dim))
    om = O.GanRnnOracle(cfg, g, d, batch_size=B, l2_scale=1e-3)
    tm = TT.GanRnnTorchTwin(cfg, g, d, l2_scale=1e-3, dtype=torch.float64)
    ol, og = om.d_tower(x, lab, ln, nr, nf)
    tl, tg = tm.d_losses_and_grads(x, lab, ln, nr, nf)
    assert np.allclose(ol, tl, rtol=1e-12)
    for k in og:
        assert np.allclose(og[k], tg[k].numpy(), rtol=1e-9, atol=1e-13), k
    ol, og, oy = om.g_tower(x, lab, ln, nf)
    tl, tg, ty = tm.g_losses_and_grads(x, lab, ln, nf)
    assert np.allclose(ol, tl, rtol=1e-12) and np.allclose(oy, ty, atol=1e-13)
    for k in og:
        assert np.allclose(og[k], tg[k].numpy(), rtol=1e-9, atol=1e-13), k


def test_no_projection_variant_matches_autograd():
    cfg = small_cfg(g_proj=0, d_proj=0)
    g, d = rand_params(cfg, seed=7)
    rng = np.random.default_rng(8)
    x = rng.normal(size=(3, 5, 9)); lab = rng.normal(size=(3, 5, 5)); ln = np.array([5, 2, 4])
    om = O.GanRnnOracle(cfg, g, d, batch_size=3)
    tm = TT.GanRnnTorchTwin(cfg, g, d, dtype=torch.float64)
    ol, og, _ = om.g_tower(x, lab, ln)
    tl, tg, _ = tm.g_losses_and_grads(x, lab, ln)
    assert np.allclose(ol, tl, rtol=1e-12)
    for k in og:
        assert np.allclose(og[k], tg[k].numpy(), rtol=1e-9, atol=1e-13), k


def test_finite_difference_g_loss():
    cfg = small_cfg()
    g, d = rand_params(cfg, seed=9)
    rng = np.random.default_rng(10)
    x = rng.normal(size=(2, 4, 9)); lab = rng.normal(size=(2, 4, 5)); ln = np.array([4, 2])
    om = O.GanRnnOracle(cfg, g, d, batch_size=2)
    _, grads, _ = om.g_tower(x, lab, ln)
    for name in ["g_model/rnn/multi_rnn_cell/cell_0/lstm_cell/kernel",
                 "g_model/rnn/multi_rnn_cell/cell_1/lstm_cell/w_o_diag",
                 "g_model/fully_connected/weights"]:
        idx = tuple(int(rng.integers(0, s)) for s in om.g[name].shape)
        eps = 1e-6
        om.g[name][idx] += eps
        lp = om.g_tower(x, lab, ln, want_grads=False)[0][3]
        om.g[name][idx] -= 2 * eps
        lm = om.g_tower(x, lab, ln, want_grads=False)[0][3]
        om.g[name][idx] += eps
        assert math.isclose((lp - lm) / (2 * eps), grads[name][idx], rel_tol=1e-5, abs_tol=1e-9), name


def test_steps_match_twin_and_towers_match_single():
    """1 D + 2 G updates: oracle == autograd twin; and 2 towers with lr*2 ==
    what average_gradients (utils/ops.py:343-376) prescribes."""
    cfg = small_cfg()
    g, d = rand_params(cfg, seed=11)
    rng = np.random.default_rng(12)
    B, T = 4, 6
    x = rng.normal(size=(B, T, 9)); lab = rng.normal(size=(B, T, 5)); ln = np.array([6, 6, 3, 2])
    om = O.GanRnnOracle(cfg, g, d, batch_size=B)
    tm = TT.GanRnnTorchTwin(cfg, g, d, dtype=torch.float64)
    a = om.d_step(x, lab, ln); b = tm.d_step(x, lab, ln)
    assert np.allclose([v[0] for v in a], b, rtol=1e-12)
    for _ in range(2):
        a = om.g_step(x, lab, ln); b = tm.g_step(x, lab, ln)
        assert np.allclose([v[0] for v in a], b, rtol=1e-10)
    for k in om.g:
        assert np.allclose(om.g[k], tm.g[k].detach().numpy(), rtol=1e-8, atol=1e-12), k
    for k in om.d:
        assert np.allclose(om.d[k], tm.d[k].detach().numpy(), rtol=1e-8, atol=1e-12), k
    # towers: mean of per-tower grads
    o2 = O.GanRnnOracle(cfg, g, d, batch_size=2, num_towers=2)
    rl, fk, dl = o2.d_step(x, lab, ln)
    assert len(rl) == 2
    o1a = O.GanRnnOracle(cfg, g, d, batch_size=2); o1b = O.GanRnnOracle(cfg, g, d, batch_size=2)
    (_, ga), (_, gb) = o1a.d_tower(x[:2], lab[:2], ln[:2]), o1b.d_tower(x[2:], lab[2:], ln[2:])
    for k in o2.d:
        want = d[k] - 1e-3 * O.clip_by_norm(0.5 * (ga[k] + gb[k]), 15.0)
        assert np.allclose(o2.d[k], want, rtol=1e-12, atol=1e-15), k


def test_train_one_iteration_skips_short_batches():
    cfg = small_cfg()
    g, d = rand_params(cfg, seed=13)
    rng = np.random.default_rng(14)
    mk = lambda b: (rng.normal(size=(b, 5, 9)), rng.normal(size=(b, 5, 5)), np.full(b, 5))
    batches = [mk(2), mk(1), mk(2)]
    om = O.GanRnnOracle(cfg, g, d, batch_size=2)
    r = O.train_one_iteration(om, batches, disc_updates=1, gen_updates=2)
    assert len(r) == 7 and om.adam_t == 4
    assert math.isclose(r[2], r[0] + r[1], rel_tol=1e-12)


def test_sequence_model_with_discriminator_dnn_matches_autograd():
    """models/discriminator_dnn.py as the D of the sequence model (the combination BASELINE.json names)."""
    cfg = small_cfg(d_type="dnn", d_layers=2, d_cells=11)
    g, d = rand_params(cfg, seed=21)
    for k in d:
        if k.endswith("weights"):
            d[k] = d[k] * 3.0                       # push some logits outside [-0.5, 1.5]
    rng = np.random.default_rng(22)
    x = rng.normal(size=(3, 5, 9)); lab = rng.normal(size=(3, 5, 5)); ln = np.array([5, 3, 4])
    om = O.GanRnnOracle(cfg, g, d, batch_size=3)
    tm = TT.GanRnnTorchTwin(cfg, g, d, dtype=torch.float64)
    ol, og = om.d_tower(x, lab, ln)
    tl, tg = tm.d_losses_and_grads(x, lab, ln)
    assert np.allclose(ol, tl, rtol=1e-12)
    for k in og:
        assert np.allclose(og[k], tg[k].numpy(), rtol=1e-9, atol=1e-13), k
    ol, og, _ = om.g_tower(x, lab, ln)
    tl, tg, _ = tm.g_losses_and_grads(x, lab, ln)
    assert np.allclose(ol, tl, rtol=1e-12)
    for k in og:
        assert np.allclose(og[k], tg[k].numpy(), rtol=1e-9, atol=1e-13), k


def test_supervised_tower_is_the_mse_gradient():
    """RNNTrainer graph (models/rnn_trainer.py:146-156): with the discriminator dropped, the generator gradient is the
    derivative of 0.5*Dout*mse alone -- central differences on a few coordinates of the fp64 oracle."""
    from tests.helpers import rand_batch, rand_params, small_cfg
    cfg = small_cfg("lstm")
    g, d = rand_params(cfg, 2, dtype=np.float64)
    o = O.GanRnnOracle(cfg, g, d, batch_size=3, mse_lambda=1.0)
    o.supervised = True
    x, lab, ln = rand_batch(cfg, 3, 5, seed=9, ragged=True)
    losses, grads, _ = o.g_tower(x, lab, ln)
    assert losses[0] == 0.0 and abs(losses[3] - losses[1]) < 1e-12
    rng = np.random.default_rng(0)
    for name in list(o.g)[::3]:
        idx = tuple(rng.integers(0, s) for s in o.g[name].shape)
        old = o.g[name][idx]
        eps = 1e-6
        o.g[name][idx] = old + eps; lp = o.g_tower(x, lab, ln, want_grads=False)[0][3]
        o.g[name][idx] = old - eps; lm = o.g_tower(x, lab, ln, want_grads=False)[0][3]
        o.g[name][idx] = old
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - grads[name][idx]) < 1e-6 * max(1.0, abs(fd)), (name, fd, grads[name][idx])


@pytest.mark.parametrize("g_type", ["lstm", "res_lstm_l"])
def test_dropout_wrapper_gradients_by_finite_differences(g_type):
    """DropoutWrapper(cell, output_keep_prob) around every generator layer (models/lstm.py:99-102, res_lstm_l.py:96-99): the
    layer's output sequence times mask / keep feeds the layer above / the residual sum / the output FC; the carried state is not
    dropped.  Masks are an input of the oracle; its hand-written backward is pinned by central differences in fp64."""
    cfg = small_cfg(g_type)
    g, _ = rand_params(cfg, seed=11)
    rng = np.random.default_rng(12)
    B, T, keep = 3, 5, 0.7
    x = rng.normal(size=(B, T, cfg.input_dim)); ln = np.array([5, 3, 1], np.int32)
    masks = [(rng.random((B, T, cfg.g_proj)) < keep).astype(np.float64) for _ in range(cfg.g_layers)]
    drop = (keep, lambda l: masks[l])
    Rw = rng.normal(size=(B, T, cfg.output_dim))
    y, cache = O.generator_fwd(cfg, g, x, ln, drop)
    y0, _ = O.generator_fwd(cfg, g, x, ln)
    assert not np.allclose(y, y0)
    grads = O.generator_bwd(cfg, g, cache, Rw)
    loss = lambda: float(np.sum(O.generator_fwd(cfg, g, x, ln, drop)[0] * Rw))
    checked = 0
    for name in sorted(g):
        for _ in range(2):
            idx = tuple(int(rng.integers(0, s)) for s in g[name].shape)
            eps = 1e-6
            g[name][idx] += eps; lp = loss()
            g[name][idx] -= 2 * eps; lm = loss()
            g[name][idx] += eps
            assert math.isclose((lp - lm) / (2 * eps), grads[name][idx], rel_tol=2e-5, abs_tol=1e-8), (name, idx)
            checked += 1
    assert checked >= 2 * len(g)
    # all-ones masks with keep = 1/2 double every layer output; an all-zero mask on the top layer leaves only the residual / bias path
    ones = (0.5, lambda l: np.ones((B, T, cfg.g_proj)))
    _, c1 = O.generator_fwd(cfg, g, x, ln, ones)
    _, c0 = O.generator_fwd(cfg, g, x, ln)
    first = c1["ins"][1] - (c1["ins"][0] if g_type == "res_lstm_l" else 0.0)
    plain = c0["ins"][1] - (c0["ins"][0] if g_type == "res_lstm_l" else 0.0)
    assert np.allclose(first, 2.0 * plain, rtol=1e-12, atol=1e-14)
    # the model-level switch: training runs only, a new run index per training sess.run, evaluation fetches undropped
    _, d = rand_params(cfg, seed=11)
    seen = []

    def mask_fn(run, tower, layer, b, t, p):
        seen.append((run, tower, layer))
        return np.ones((b, t, p))
    om = O.GanRnnOracle(cfg, g, d, batch_size=B, keep_prob=0.5, mask_fn=mask_fn)
    lab = rng.normal(size=(B, T, cfg.output_dim))
    ev = om.g_step(x, lab, ln, train=False)
    assert seen == [] and np.allclose(ev, O.GanRnnOracle(cfg, g, d, batch_size=B).g_step(x, lab, ln, train=False), rtol=1e-12)
    om.d_step(x, lab, ln); om.g_step(x, lab, ln)
    assert sorted(set(seen)) == [(r, 0, l) for r in (1, 2) for l in range(cfg.g_layers)]
