"""Frame-level DNN-GAN (models/gan.py + dnn.py + discriminator_dnn.py) and discriminator_dnn as the D of the
sequence model: HIP path through the C ABI vs the fp64 oracles."""
from types import SimpleNamespace

import numpy as np
import pytest

from oracle import dnn_gan_oracle as DO
from oracle import rsrgan_oracle as O
from tests.helpers import NET_D, NET_G, args_for, dropout_mask, overrides, rand_batch, rand_params, rel_err, small_cfg, split_flat

pytestmark = pytest.mark.gpu


def _drop_kw(kw):
    """the oracle takes the dropout masks as an input: the ones the device draws (GAN's default seed 4321, rank 0)"""
    keep = kw.get("keep_prob", 1.0)
    if keep >= 1.0:
        return {}
    return dict(keep_prob=float(np.float32(keep)),
                mask_fn=lambda run, net, layer, call, rows, cols: dropout_mask(4321, run, net, layer, call, rows, cols, keep))


def _dnn_pair(cfg, N, seed, **kw):
    from rsrgan_amd import GAN
    rng = np.random.default_rng(seed)
    g = {k: v.astype(np.float32) for k, v in DO.init_params(DO.g_param_specs(cfg), rng).items()}
    d = {k: (v * 2.0).astype(np.float32) for k, v in DO.init_params(DO.d_param_specs(cfg), rng, relu_init=True).items()}
    for p in (g, d):
        for k in p:
            if k.endswith("biases"):
                p[k] = rng.normal(0, 0.1, p[k].shape).astype(np.float32)
    args = SimpleNamespace(batch_size=N, input_dim=cfg.input_dim, output_dim=cfg.output_dim, left_context=cfg.left_context,
                           right_context=cfg.right_context, g_type="dnn", keep_prob=kw.get("keep_prob", 1.0), batch_norm=False, num_gpu=1,
                           save_dir=None, l2_scale=kw.get("l2_scale", 0.0), disc_updates=1, gen_updates=1, init_mse_weight=10.0,
                           d_learning_rate=kw.get("d_lr", 1e-4), g_learning_rate=kw.get("g_lr", 1e-4))
    m = GAN(None, args, ["gpu:0"], net_overrides=dict(g_layers=cfg.g_hidden, g_cells=cfg.g_units, d_layers=cfg.d_hidden, d_cells=cfg.d_units))
    assert [(n, tuple(s)) for n, s, _ in m.engine.tensor_table(NET_G)] == [(n, tuple(s)) for n, s in DO.g_param_specs(cfg)]
    assert [(n, tuple(s)) for n, s, _ in m.engine.tensor_table(NET_D)] == [(n, tuple(s)) for n, s in DO.d_param_specs(cfg)]
    m.set_vars(g, d)
    o = DO.GanDnnOracle(cfg, g, d, l2_scale=kw.get("l2_scale", 0.0), g_learning_rate=float(np.float32(kw.get("g_lr", 1e-4))),
                        d_learning_rate=float(np.float32(kw.get("d_lr", 1e-4))), **_drop_kw(kw))
    return m, o


@pytest.mark.parametrize("N", [7, 130])
def test_frame_level_gan_towers_and_steps(N):
    cfg = DO.DnnCfg(input_dim=6, output_dim=5, left_context=2, right_context=1, g_units=20, g_hidden=3, d_units=18, d_hidden=2)
    m, o = _dnn_pair(cfg, N, seed=N, l2_scale=1e-3, g_lr=1e-3, d_lr=2e-3)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((N, cfg.fed_dim)).astype(np.float32); lab = rng.standard_normal((N, cfg.output_dim)).astype(np.float32)
    y = m.forward(x)
    assert np.abs(y - o.forward(x)).max() < 1e-4
    got = m.engine.d_backward(x[:, None], lab[:, None], None, train=True, apply=False).cpu().numpy()
    want, wg = o.d_tower(x, lab)
    assert np.allclose(got, want, rtol=1e-4), (got, want)
    gr = split_flat(m.engine.get_grads(NET_D).cpu().numpy(), m.engine.tensor_table(NET_D))
    for k in wg:
        assert rel_err(gr[k], wg[k]) < 2e-3, k
    got = m.engine.g_backward(x[:, None], lab[:, None], None, train=True, reuse=True, apply=False).cpu().numpy()
    want, wg, _ = o.g_tower(x, lab)
    assert np.allclose(got, want, rtol=1e-4), (got, want)
    gr = split_flat(m.engine.get_grads(NET_G).cpu().numpy(), m.engine.tensor_table(NET_G))
    for k in wg:
        assert rel_err(gr[k], wg[k]) < 2e-3, k
    # 1 D + 2 G updates: Adam on both nets, no clipping
    assert np.allclose(np.ravel(m.d_step(x, lab)), o.d_step(x, lab), rtol=1e-4)
    assert np.allclose(np.ravel(m.g_step(x, lab, reuse_g_forward=True)), o.g_step(x, lab), rtol=1e-4)
    assert np.allclose(np.ravel(m.g_step(x, lab)), o.g_step(x, lab), rtol=1e-4)
    gv, dv = m.get_vars()
    for k in o.g:
        assert rel_err(gv[k], o.g[k]) < 1e-4, k
    for k in o.d:
        assert rel_err(dv[k], o.d[k]) < 1e-4, k
    assert m.engine.get_scalar("adam_step") == 2 and m.engine.get_scalar("adam_step_d") == 1
    ev = np.ravel(m.g_step(x, lab, train=False))          # train=0 == the cross_validation twin's fetch: no L2 term (gan.py:207)
    o.cross_validation = True
    assert np.allclose(ev, o.g_step(x, lab, train=False), rtol=1e-4) and ev[2] == 0.0


def test_frame_level_gan_reference_sizes():
    """models/dnn.py / discriminator_dnn.py hard-coded sizes: 2827 -> 4x1024 -> 40 ; 297 -> 4x1024 -> 1."""
    cfg = DO.DnnCfg()
    N = 96
    m, o = _dnn_pair(cfg, N, seed=5)
    rng = np.random.default_rng(6)
    x = rng.standard_normal((N, cfg.fed_dim)).astype(np.float32); lab = rng.standard_normal((N, cfg.output_dim)).astype(np.float32)
    assert np.allclose(np.ravel(m.d_step(x, lab, train=False)), o.d_step(x, lab, train=False), rtol=1e-4)
    got = m.engine.g_backward(x[:, None], lab[:, None], None, train=True, reuse=False, apply=False).cpu().numpy()
    want, wg, y = o.g_tower(x, lab)
    assert np.allclose(got, want, rtol=1e-4), (got, want)
    gr = split_flat(m.engine.get_grads(NET_G).cpu().numpy(), m.engine.tensor_table(NET_G))
    for k in wg:
        assert rel_err(gr[k], wg[k]) < 2e-3, k
    assert np.abs(m.forward(x) - y).mean() / np.abs(y).mean() < 1e-4


@pytest.mark.parametrize("flags", [0, 1])
def test_sequence_model_with_discriminator_dnn(flags):
    from rsrgan_amd import GAN_RNN
    cfg = small_cfg("lstm", d_type="dnn", d_layers=2, d_cells=11)
    B, T = 5, 6
    g, d = rand_params(cfg, 7)
    for k in d:
        if k.endswith("weights"):
            d[k] = (d[k] * 3.0).astype(np.float32)
    m = GAN_RNN(None, args_for(cfg, B), ["gpu:0"], max_frames=T,
                net_overrides=dict(g_layers=cfg.g_layers, g_cells=cfg.g_cells, g_proj=cfg.g_proj, d_type="dnn", d_layers=2, d_cells=11, flags=flags))
    assert [(n, tuple(s)) for n, s, _ in m.engine.tensor_table(NET_D)] == [(n, tuple(s)) for n, s in O.d_param_specs(cfg)]
    m.set_vars(g, d)
    o = O.GanRnnOracle(cfg, g, d, batch_size=B, g_learning_rate=float(np.float32(8e-5)), d_learning_rate=float(np.float32(1e-3)))
    x, lab, ln = rand_batch(cfg, B, T, 8, ragged=True)
    got = m.engine.d_backward(x, lab, ln, train=True, apply=False).cpu().numpy()
    want, wg = o.d_tower(x.astype(np.float64), lab.astype(np.float64), ln)
    assert np.allclose(got, want, rtol=1e-4), (got, want)
    gr = split_flat(m.engine.get_grads(NET_D).cpu().numpy(), m.engine.tensor_table(NET_D))
    for k in wg:
        assert rel_err(gr[k], wg[k]) < 2e-3, k
    got = m.engine.g_backward(x, lab, ln, train=True, reuse=True, apply=False).cpu().numpy()
    want, wg, _ = o.g_tower(x.astype(np.float64), lab.astype(np.float64), ln)
    assert np.allclose(got, want, rtol=1e-4), (got, want)
    gr = split_flat(m.engine.get_grads(NET_G).cpu().numpy(), m.engine.tensor_table(NET_G))
    for k in wg:
        assert rel_err(gr[k], wg[k]) < 2e-3 or np.abs(gr[k] - wg[k]).max() < 1e-7, k
    assert np.allclose(np.ravel(m.d_step(x, lab, ln)), np.ravel(o.d_step(x, lab, ln)), rtol=1e-4)      # SGD + clip for D here
    assert np.allclose(np.ravel(m.g_step(x, lab, ln)), np.ravel(o.g_step(x, lab, ln)), rtol=1e-4)
    _, dv = m.get_vars()
    for k in o.d:
        assert rel_err(dv[k], o.d[k]) < 1e-4, k


def _bn_pair(cfg, N, seed, cross_validation=False, g=None, d=None, **kw):
    from rsrgan_amd import GAN
    rng = np.random.default_rng(seed)
    if g is None:
        g = DO.init_params(DO.g_param_specs(cfg), rng)
        d = {k: (v * 2.0 if k.endswith("weights") else v) for k, v in DO.init_params(DO.d_param_specs(cfg), rng, relu_init=True).items()}
        for p in (g, d):                     # a state some way into training: r != 1, d != 0
            w = rng.uniform(0.2, 0.6)
            for k in p:
                if k.endswith("biases") or k.endswith("/beta"):
                    p[k] = rng.normal(0, 0.1, p[k].shape)
                elif k.endswith("/gamma"):
                    p[k] = rng.uniform(0.7, 1.3, p[k].shape)
                elif k.endswith("renorm_mean"):
                    p[k] = w * rng.normal(0, 0.3, p[k].shape)
                elif k.endswith("renorm_stddev"):
                    p[k] = w * rng.uniform(0.5, 1.5, p[k].shape)
                elif k.endswith("_weight"):
                    p[k] = np.float64(w)
                elif k.endswith("moving_mean"):
                    p[k] = rng.normal(0, 0.3, p[k].shape)
                elif k.endswith("moving_variance"):
                    p[k] = rng.uniform(0.5, 1.5, p[k].shape)
    g = {k: np.asarray(v, np.float32) for k, v in g.items()}
    d = {k: np.asarray(v, np.float32) for k, v in d.items()}
    args = SimpleNamespace(batch_size=N, input_dim=cfg.input_dim, output_dim=cfg.output_dim, left_context=cfg.left_context,
                           right_context=cfg.right_context, g_type="dnn", keep_prob=kw.get("keep_prob", 1.0), batch_norm=True, num_gpu=1,
                           save_dir=None, l2_scale=kw.get("l2_scale", 0.0), disc_updates=1, gen_updates=1, init_mse_weight=10.0,
                           d_learning_rate=kw.get("d_lr", 1e-4), g_learning_rate=kw.get("g_lr", 1e-4))
    m = GAN(None, args, ["gpu:0"], cross_validation=cross_validation,
            net_overrides=dict(g_layers=cfg.g_hidden, g_cells=cfg.g_units, d_layers=cfg.d_hidden, d_cells=cfg.d_units))
    shape1 = lambda s: (1,) if tuple(s) == () else tuple(s)       # the ABI reports a scalar variable as one element
    assert [(n, tuple(s)) for n, s, _ in m.engine.tensor_table(NET_G)] == [(n, shape1(s)) for n, s in DO.g_param_specs(cfg)]
    assert [(n, tuple(s)) for n, s, _ in m.engine.tensor_table(NET_D)] == [(n, shape1(s)) for n, s in DO.d_param_specs(cfg)]
    m.set_vars(g, d)
    o = DO.GanDnnOracle(cfg, g, d, l2_scale=kw.get("l2_scale", 0.0), g_learning_rate=float(np.float32(kw.get("g_lr", 1e-4))),
                        d_learning_rate=float(np.float32(kw.get("d_lr", 1e-4))), cross_validation=cross_validation, **_drop_kw(kw))
    return m, o


def _cmp_vars(m, o, tol=2e-4):
    gv, dv = m.get_vars()
    for got, want in ((gv, o.g), (dv, o.d)):
        for k in want:
            assert rel_err(got[k], want[k]) < tol or np.abs(got[k] - want[k]).max() < 1e-6, (k, got[k], want[k])


@pytest.mark.parametrize("N", [8, 130])
def test_frame_level_gan_batch_norm_renorm(N):
    """run_gan_dnn.sh:134 --batch_norm=true: relu(batch_norm(x.W, scale=True, renorm=True)) in G and D (dnn.py:56-61,
    discriminator_dnn.py:36-41) against oracle/bn_renorm.py: towers, gradients, update ops, Adam steps, the cross_validation twin."""
    cfg = DO.DnnCfg(input_dim=6, output_dim=5, left_context=2, right_context=1, g_units=20, g_hidden=3, d_units=24, d_hidden=2, batch_norm=True)
    m, o = _bn_pair(cfg, N, seed=N, l2_scale=1e-3, g_lr=1e-3, d_lr=2e-3)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((N, cfg.fed_dim)).astype(np.float32); lab = rng.standard_normal((N, cfg.output_dim)).astype(np.float32)
    y = m.forward(x)                                        # is_training=True graph: batch statistics, no update ops fetched
    assert np.abs(y - o.forward(x)).max() < 2e-4
    _cmp_vars(m, o)
    got = m.engine.d_backward(x[:, None], lab[:, None], None, train=True, apply=False).cpu().numpy()
    want, wg = o.d_tower(x, lab)
    assert np.allclose(got, want, rtol=2e-4), (got, want)
    gr = split_flat(m.engine.get_grads(NET_D).cpu().numpy(), m.engine.tensor_table(NET_D))
    for k in wg:
        assert rel_err(gr[k], wg[k]) < 2e-3, k
    _cmp_vars(m, o)                                         # the run's update ops: G x2, D real x2, D fake x1
    got = m.engine.g_backward(x[:, None], lab[:, None], None, train=True, reuse=True, apply=False).cpu().numpy()
    want, wg, _ = o.g_tower(x, lab)
    assert np.allclose(got, want, rtol=2e-4), (got, want)
    gr = split_flat(m.engine.get_grads(NET_G).cpu().numpy(), m.engine.tensor_table(NET_G))
    for k in wg:
        assert rel_err(gr[k], wg[k]) < 2e-3, k
    _cmp_vars(m, o)
    x2 = rng.standard_normal((N, cfg.fed_dim)).astype(np.float32) * 1.5 + 0.3
    assert np.allclose(np.ravel(m.d_step(x2, lab)), o.d_step(x2, lab), rtol=2e-4)
    assert np.allclose(np.ravel(m.g_step(x2, lab, reuse_g_forward=True)), o.g_step(x2, lab), rtol=2e-4)
    assert np.allclose(np.ravel(m.g_step(x, lab)), o.g_step(x, lab), rtol=2e-4)
    _cmp_vars(m, o)
    for k, v in m.get_vars()[0].items():                    # the statistics are no trainable variables: Adam leaves them alone
        if k.endswith("renorm_mean_weight"):
            assert np.isclose(float(v), float(o.g[k]), rtol=1e-5)
    # train=False on the training model IS the cross_validation twin's fetch on the shared variables: moving statistics, no L2, no update
    assert np.allclose(np.ravel(m.d_step(x2, lab, train=False)), o.d_step(x2, lab, train=False), rtol=2e-4)
    ev = np.ravel(m.g_step(x2, lab, train=False))
    assert np.allclose(ev, o.g_step(x2, lab, train=False), rtol=2e-4) and ev[2] == 0.0
    _cmp_vars(m, o)
    # the cross_validation twin shares the variables and normalises with the moving statistics (is_training=False)
    mcv, ocv = _bn_pair(cfg, N, seed=0, cross_validation=True, g=o.g, d=o.d)
    assert np.allclose(np.ravel(mcv.g_step(x2, lab, train=False)), ev, rtol=1e-5)
    assert np.allclose(np.ravel(mcv.d_step(x, lab, train=False)), ocv.d_step(x, lab, train=False), rtol=2e-4)
    assert np.allclose(np.ravel(mcv.g_step(x, lab, train=False)), ocv.g_step(x, lab, train=False), rtol=2e-4)
    assert np.abs(mcv.forward(x) - ocv.forward(x)).max() < 2e-4
    _cmp_vars(mcv, ocv)


def test_frame_level_gan_batch_norm_reference_sizes():
    """2827 -> 4 x [1024, BN, ReLU] -> 40 ; 297 -> 4 x [1024, BN, ReLU] -> 1 at the shipped batch size (run_gan_dnn.sh:124: 256)."""
    cfg = DO.DnnCfg(batch_norm=True)
    N = 256
    m, o = _bn_pair(cfg, N, seed=5)
    rng = np.random.default_rng(6)
    x = rng.standard_normal((N, cfg.fed_dim)).astype(np.float32); lab = rng.standard_normal((N, cfg.output_dim)).astype(np.float32)
    got = m.engine.d_backward(x[:, None], lab[:, None], None, train=True, apply=False).cpu().numpy()
    want, wg = o.d_tower(x, lab)
    assert np.allclose(got, want, rtol=2e-4), (got, want)
    gr = split_flat(m.engine.get_grads(NET_D).cpu().numpy(), m.engine.tensor_table(NET_D))
    for k in wg:
        assert rel_err(gr[k], wg[k]) < 3e-3, k
    got = m.engine.g_backward(x[:, None], lab[:, None], None, train=True, reuse=False, apply=False).cpu().numpy()
    want, wg, y = o.g_tower(x, lab)
    assert np.allclose(got, want, rtol=2e-4), (got, want)
    gr = split_flat(m.engine.get_grads(NET_G).cpu().numpy(), m.engine.tensor_table(NET_G))
    for k in wg:
        assert rel_err(gr[k], wg[k]) < 3e-3, k
    _cmp_vars(m, o)


@pytest.mark.parametrize("N,batch_norm", [(7, False), (130, False), (64, True)])
def test_frame_level_gan_dropout(N, batch_norm):
    """--keep_prob < 1 (scripts/train_gan_dnn.py): tf.nn.dropout after every hidden ReLU of G and D (dnn.py:86,99,
    discriminator_dnn.py:68,81), new masks in every training run, the D's real and fake calls with their own; the oracle is fed
    the masks the device draws (tests/helpers.py dropout_mask).  Towers, every gradient, Adam steps; evaluation runs and a model
    with l2_scale = 0 do not drop (dnn.py:67-71)."""
    cfg = DO.DnnCfg(input_dim=6, output_dim=5, left_context=2, right_context=1, g_units=20, g_hidden=3, d_units=24, d_hidden=2,
                    batch_norm=batch_norm)
    pair = _bn_pair if batch_norm else _dnn_pair
    m, o = pair(cfg, N, seed=N, l2_scale=1e-3, g_lr=1e-3, d_lr=2e-3, keep_prob=0.8)
    assert m.keep_prob == 0.8
    rng = np.random.default_rng(3)
    x = rng.standard_normal((N, cfg.fed_dim)).astype(np.float32); lab = rng.standard_normal((N, cfg.output_dim)).astype(np.float32)
    got = m.engine.d_backward(x[:, None], lab[:, None], None, train=True, apply=False).cpu().numpy()
    want, wg = o.d_tower(x, lab)
    assert np.allclose(got, want, rtol=2e-4), (got, want)
    plain = pair(cfg, N, seed=N, l2_scale=1e-3, g_lr=1e-3, d_lr=2e-3)[1].d_tower(x, lab)[0]
    assert not np.allclose(want, plain, rtol=1e-3)                  # (the masks do change the losses)
    gr = split_flat(m.engine.get_grads(NET_D).cpu().numpy(), m.engine.tensor_table(NET_D))
    for k in wg:
        assert rel_err(gr[k], wg[k]) < 2e-3, k
    got = m.engine.g_backward(x[:, None], lab[:, None], None, train=True, reuse=True, apply=False).cpu().numpy()      # reuse is ignored: new masks
    want, wg, _ = o.g_tower(x, lab)
    assert np.allclose(got, want, rtol=2e-4), (got, want)
    gr = split_flat(m.engine.get_grads(NET_G).cpu().numpy(), m.engine.tensor_table(NET_G))
    for k in wg:
        assert rel_err(gr[k], wg[k]) < 2e-3, k
    for _ in range(2):
        assert np.allclose(np.ravel(m.d_step(x, lab)), o.d_step(x, lab), rtol=2e-4)
        assert np.allclose(np.ravel(m.g_step(x, lab, reuse_g_forward=True)), o.g_step(x, lab), rtol=2e-4)
    _cmp_vars(m, o)
    # evaluation fetches (the cross_validation twin's) and inference do not drop
    assert np.allclose(np.ravel(m.d_step(x, lab, train=False)), o.d_step(x, lab, train=False), rtol=2e-4)
    assert np.allclose(np.ravel(m.g_step(x, lab, train=False)), o.g_step(x, lab, train=False), rtol=2e-4)
    if not batch_norm:
        assert np.abs(m.forward(x) - o.forward(x)).max() < 2e-4
    # l2_scale = 0: the reference resets keep_prob to 1.0
    m0, o0 = pair(cfg, N, seed=N, l2_scale=0.0, keep_prob=0.8)
    assert m0.keep_prob == 1.0
    assert np.allclose(np.ravel(m0.d_step(x, lab)), o0.d_step(x, lab), rtol=2e-4)
