"""Generator-only trainers (models/rnn_trainer.py, models/dnn_trainer.py) on the HIP path vs the fp64 oracles
in supervised mode: g_loss = 0.5*Dout*mse + l2, Adam, clip 15 (RNN) / none (DNN)."""
from types import SimpleNamespace

import os

import numpy as np
import pytest

from oracle import dnn_gan_oracle as DO
from oracle import rsrgan_oracle as O
from tests.helpers import NET_D, NET_G, args_for, overrides, rand_batch, rand_params, rel_err, small_cfg, split_flat

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("flags", [0, 1])
@pytest.mark.parametrize("g_type", ["lstm", "res_lstm_l", "res_lstm_base"])
def test_rnn_trainer_matches_oracle(g_type, flags):
    from rsrgan_amd.trainer import RNNTrainer
    cfg = small_cfg(g_type)
    B, T = 5, 7
    g, d = rand_params(cfg, 11)
    args = args_for(cfg, B, l2_scale=1e-3, g_learning_rate=1e-3)
    m = RNNTrainer(None, args, ["gpu:0"], max_frames=T, net_overrides=dict(overrides(cfg), flags=flags))
    m.set_vars(g, d)
    o = O.GanRnnOracle(cfg, g, d, batch_size=B, l2_scale=1e-3, g_learning_rate=float(np.float32(1e-3)), mse_lambda=1.0)
    o.supervised = True
    x, lab, ln = rand_batch(cfg, B, T, seed=5, ragged=True)
    # tower: losses and gradients
    got = m.engine.g_backward(x, lab, ln, None, train=True, apply=False).cpu().numpy()
    want, wg, _ = o.g_tower(x, lab, ln)
    assert got[0] == 0.0 and np.allclose(got[1:], want[1:], rtol=1e-4), (got, want)
    gr = split_flat(m.engine.get_grads(NET_G).cpu().numpy(), m.engine.tensor_table(NET_G))
    for k in wg:
        assert rel_err(gr[k], wg[k]) < 2e-3, k
    # three optimizer steps, then the variables
    for i in range(3):
        xs, ls, lns = rand_batch(cfg, B, T, seed=20 + i, ragged=i % 2 == 0)
        got = np.ravel(m.step(xs, ls, lns))
        w = o.g_step(xs, ls, lns)
        assert np.allclose(got, np.ravel(w)[1:], rtol=2e-4), (i, got, w)
    gv, _ = m.get_vars()
    for k in o.g:
        assert rel_err(gv[k], o.g[k]) < 1e-3, k
    with pytest.raises(RuntimeError):
        m.d_step(x, lab, ln)
    from rsrgan_amd._lib import RsrganError
    with pytest.raises(RsrganError):
        m.engine.d_backward(x, lab, ln)
    # eval fetch on the cross-validation twin: no l2 term
    o.cross_validation = True
    ev = np.ravel(RNNTrainer(None, args, ["gpu:0"], cross_validation=True, share_engine_from=m).step(x, lab, ln))
    w = o.g_step(x, lab, ln, train=False)
    assert np.allclose(ev, np.ravel(w)[1:], rtol=2e-4) and ev[1] == 0.0


@pytest.mark.parametrize("g_type,flags", [("lstm", 1), ("res_lstm_base", 0)])
def test_rnn_trainer_with_dropout_wrapper(g_type, flags):
    """models/rnn_trainer.py:76-78 keep_prob + the generators' DropoutWrapper (lstm.py:99-102, res_lstm_base.py:96-99) on the
    supervised path: three Adam steps with the device's masks fed to the oracle, then every variable; the evaluation fetch
    is undropped."""
    from rsrgan_amd.trainer import RNNTrainer
    from tests.helpers import seq_dropout_mask
    cfg = small_cfg(g_type)
    B, T, keep = 5, 7, 0.75
    g, d = rand_params(cfg, 11)
    args = args_for(cfg, B, l2_scale=1e-3, g_learning_rate=1e-3, keep_prob=keep)
    m = RNNTrainer(None, args, ["gpu:0"], max_frames=T, net_overrides=dict(overrides(cfg), flags=flags))
    m.set_vars(g, d)
    o = O.GanRnnOracle(cfg, g, d, batch_size=B, l2_scale=1e-3, g_learning_rate=float(np.float32(1e-3)), mse_lambda=1.0,
                       keep_prob=float(np.float32(keep)),
                       mask_fn=lambda run, tower, layer, b, t, p: seq_dropout_mask(4321, run, layer, b, t, p, keep))
    o.supervised = True
    plain = O.GanRnnOracle(cfg, g, d, batch_size=B, l2_scale=1e-3, g_learning_rate=float(np.float32(1e-3)), mse_lambda=1.0)
    plain.supervised = True
    for i in range(3):
        xs, ls, lns = rand_batch(cfg, B, T, seed=20 + i, ragged=i % 2 == 0)
        got = np.ravel(m.step(xs, ls, lns))
        w = o.g_step(xs, ls, lns)
        assert np.allclose(got, np.ravel(w)[1:], rtol=2e-4), (i, got, w)
        if i == 0:
            assert not np.allclose(np.ravel(w)[1], np.ravel(plain.g_step(xs, ls, lns))[1], rtol=1e-3)
    gv, _ = m.get_vars()
    for k in o.g:
        assert rel_err(gv[k], o.g[k]) < 1e-3, k
    x, lab, ln = rand_batch(cfg, B, T, seed=5, ragged=True)
    ev = np.ravel(m.step(x, lab, ln, train=False)); w = np.ravel(o.g_step(x, lab, ln, train=False))
    assert np.allclose(ev[0], w[1], rtol=2e-4)


DNN_TRAINER_CASES = {
    "toy-9": (9, dict(input_dim=6, output_dim=5, left_context=2, right_context=1, g_units=20, g_hidden=3, d_units=18, d_hidden=2)),
    "toy-200": (200, dict(input_dim=6, output_dim=5, left_context=2, right_context=1, g_units=20, g_hidden=3, d_units=18, d_hidden=2)),
    # BASELINE.json configs[0] at its own size: models/dnn.py:32-114 (1 + 3 hidden ReLU-1024 layers + linear 40) on 1000 frames of
    # 257-dim LPS, context 0 (the shape bench.py --net dnn_trainer --batch 1000 times)
    "configs0-1000x257": (1000, dict(input_dim=257, output_dim=40, left_context=0, right_context=0, g_units=1024, g_hidden=4, d_units=18, d_hidden=2)),
}


@pytest.mark.parametrize("case", sorted(DNN_TRAINER_CASES))
def test_dnn_trainer_matches_oracle(case):
    from rsrgan_amd.trainer import DNNTrainer
    N, kw = DNN_TRAINER_CASES[case]
    cfg = DO.DnnCfg(**kw)
    rng = np.random.default_rng(N)
    g = {k: v.astype(np.float32) for k, v in DO.init_params(DO.g_param_specs(cfg), rng).items()}
    d = {k: v.astype(np.float32) for k, v in DO.init_params(DO.d_param_specs(cfg), rng, relu_init=True).items()}
    args = SimpleNamespace(batch_size=N, input_dim=cfg.input_dim, output_dim=cfg.output_dim, left_context=cfg.left_context,
                           right_context=cfg.right_context, g_type="dnn", keep_prob=1.0, batch_norm=False, num_gpu=1,
                           save_dir=None, l2_scale=1e-3, g_learning_rate=1e-3)
    m = DNNTrainer(None, args, ["gpu:0"], net_overrides=dict(g_layers=cfg.g_hidden, g_cells=cfg.g_units, d_layers=cfg.d_hidden, d_cells=cfg.d_units))
    m.set_vars(g, d)
    o = DO.GanDnnOracle(cfg, g, d, l2_scale=1e-3, g_learning_rate=float(np.float32(1e-3)), mse_lambda=1.0)
    o.supervised = True
    x = rng.standard_normal((N, cfg.fed_dim)).astype(np.float32); lab = rng.standard_normal((N, cfg.output_dim)).astype(np.float32)
    got = m.engine.g_backward(x[:, None], lab[:, None], None, train=True, apply=False).cpu().numpy()
    want, wg, _ = o.g_tower(x, lab)
    assert got[0] == 0.0 and np.allclose(got[1:], want[1:], rtol=1e-4), (got, want)
    gr = split_flat(m.engine.get_grads(NET_G).cpu().numpy(), m.engine.tensor_table(NET_G))
    for k in wg:
        assert rel_err(gr[k], wg[k]) < 2e-3, k
    for i in range(3):
        got = np.ravel(m.step(x, lab))
        assert np.allclose(got, np.ravel(o.g_step(x, lab))[1:], rtol=2e-4)
    gv, _ = m.get_vars()
    for k in o.g:
        # (Adam divides by sqrt(v): an element whose gradient is a few ulps of noise still moves by ~lr per step, so after three
        # updates a small tensor such as a bias row differs by rounding-sized gradients times 1 / sqrt(v): 1.0e-3 was observed at the
        # configs[0] size only -- the toy cases keep the tighter bound)
        assert rel_err(gv[k], o.g[k]) < (2e-3 if case.startswith("configs0") else 1e-3), k


def test_training_converges_on_a_learnable_task():
    """End-to-end sanity of the whole path (forward, BPTT, clip, Adam, SGD): labels are a fixed linear map of the inputs, so
    the supervised loss must fall steadily under the RNN trainer, and under the GAN recipe (1 D + 1 G per batch, mse_lambda
    10) g_mse must fall too while every loss stays finite."""
    from rsrgan_amd import GAN_RNN, train_one_iteration
    from rsrgan_amd.trainer import RNNTrainer
    cfg = small_cfg("lstm")
    B, T = 16, 12
    rng = np.random.default_rng(0)
    A = rng.standard_normal((cfg.input_dim, cfg.output_dim)).astype(np.float32) / np.sqrt(cfg.input_dim)

    def batch(seed):
        r = np.random.default_rng(seed)
        x = r.standard_normal((B, T, cfg.input_dim)).astype(np.float32)
        ln = r.integers(T // 2, T + 1, size=B).astype(np.int32); ln[0] = T
        lab = (x @ A).astype(np.float32)
        for i in range(B):
            lab[i, ln[i]:] = 0.0
        return x, lab, ln

    args = args_for(cfg, B, g_learning_rate=3e-3, d_learning_rate=1e-3)
    tr = RNNTrainer(None, args, ["gpu:0"], max_frames=T, net_overrides=overrides(cfg))
    first = np.mean([tr.step(*batch(s))[0][0] for s in range(5)])
    for s in range(5, 300):
        tr.step(*batch(s))
    last = np.mean([tr.step(*batch(1000 + s), train=False)[0][0] for s in range(5)])
    assert np.isfinite(last) and last < 0.35 * first, (first, last)

    gan = GAN_RNN(None, args, ["gpu:0"], max_frames=T, net_overrides=overrides(cfg))
    hist = []
    for it in range(6):
        queue = [[None, *batch(2000 + 40 * it + k)] for k in range(40)]
        hist.append(train_one_iteration(None, gan, len(queue), it, queue))
    hist = np.asarray(hist)                       # columns: d_rl, d_fk, d, g_adv, g_mse, g_l2, g
    assert np.all(np.isfinite(hist))
    assert hist[-1, 4] < 0.6 * hist[0, 4], hist[:, 4]


@pytest.mark.parametrize("N,gan,ctx,width", [(6, False, (2, 1), 9), (37, False, (2, 1), 9), (6, True, (2, 1), 9),
                                              (5, False, (1, 1), 9), (5, True, (2, 2), 21), (3, False, (1, 1), 200),
                                              (3, False, (1, 1), -11), (2, True, (5, 5), -40),
                                              # BASELINE.json configs[3]'s own frame shape: 257-dim LPS x splice 11 (4 strips of 65 columns),
                                              # the reference's filter table (models/rced.py:90-114), supervised and with discriminator_dnn
                                              (2, False, (5, 5), -257), (3, True, (5, 5), -257)])     # width < 0: the reference's filter table
def test_rced_generator_matches_oracle(N, gan, ctx, width):
    """models/rced.py under DNNTrainer (and, as BASELINE.json's config 4 words it, paired with discriminator_dnn): conv2d SAME as
    patch-matrix GEMMs on the HIP path vs the fp64 oracle (tower losses, every gradient tensor, Adam steps, variables)."""
    from oracle import rced_oracle as R
    from rsrgan_amd import GAN
    from rsrgan_amd.trainer import DNNTrainer
    # even splice (2,1): patch-matrix GEMMs; odd splice: the implicit-GEMM conv (one strip per frame, or 64-column strips at width 200)
    full = width < 0
    width = abs(width)
    cfg = R.RcedCfg(input_dim=width, output_dim=5, left_context=ctx[0], right_context=ctx[1], d_units=18, d_hidden=2,
                    filters_num=R.FILTERS_NUM if full else (4,) * 9)   # the library scales the reference table 12..32..12 by g_cells/32
    rng = np.random.default_rng(N)
    g = {k: v.astype(np.float32) for k, v in R.init_params(R.g_param_specs(cfg), rng).items()}
    for k in g:
        if k.endswith("biases"):
            g[k] = rng.normal(0.05, 0.1, g[k].shape).astype(np.float32)
    if full:
        # A 257 x 11 frame has ~0.5 M ReLU units, so with random biases ~10 pre-activations per frame land within fp32 rounding
        # (1e-6) of the kink, where fp32 and the fp64 oracle legitimately pick different subgradients (seen: one unit of Conv_3
        # at |z| = 6e-8 moved that layer's gradient by 4e-3).  Here every channel sits well on one side of the kink (bias +-1,
        # conv weights x 0.1: |z| > 0.05, checked below); random ReLU masks are covered by the narrower cases above.
        for k in g:
            if "Conv" in k and k.endswith("biases"):
                g[k] = (rng.choice([-1.0, 1.0], size=g[k].shape, p=[0.25, 0.75]) * rng.uniform(0.8, 1.2, g[k].shape)).astype(np.float32)
            elif "Conv" in k and k.endswith("weights"):
                g[k] = (0.1 * g[k]).astype(np.float32)
    d = {k: (2.0 * v).astype(np.float32) for k, v in DO.init_params(DO.d_param_specs(cfg), rng, relu_init=True).items()}
    args = SimpleNamespace(batch_size=N, input_dim=cfg.input_dim, output_dim=cfg.output_dim, left_context=cfg.left_context,
                           right_context=cfg.right_context, g_type="rced", keep_prob=1.0, batch_norm=False, num_gpu=1, save_dir=None,
                           l2_scale=1e-3, g_learning_rate=1e-3, d_learning_rate=2e-3, init_mse_weight=10.0, disc_updates=1, gen_updates=1)
    ov = dict(g_layers=9, g_cells=32 if full else 4, d_layers=cfg.d_hidden, d_cells=cfg.d_units)
    if gan:
        class RcedGan(GAN):
            G_TYPES = ("dnn", "rced")
        m = RcedGan(None, args, ["gpu:0"], net_overrides=ov)
        o = R.GanRcedOracle(cfg, g, d, l2_scale=1e-3, g_learning_rate=float(np.float32(1e-3)), d_learning_rate=float(np.float32(2e-3)))
    else:
        m = DNNTrainer(None, args, ["gpu:0"], net_overrides=ov)
        o = R.GanRcedOracle(cfg, g, d, l2_scale=1e-3, g_learning_rate=float(np.float32(1e-3)), mse_lambda=1.0)
        o.supervised = True
    table = [(n, m._tf_shape(n, s)) for n, s, _ in m.engine.tensor_table(NET_G)]
    assert table == [(n, tuple(s)) for n, s in R.g_param_specs(cfg)], table
    m.set_vars(g, d)
    x = rng.standard_normal((N, cfg.fed_dim)).astype(np.float32); lab = rng.standard_normal((N, cfg.output_dim)).astype(np.float32)
    if full:
        _, (convs, _) = R.rced_fwd(cfg, {k: v.astype(np.float64) for k, v in g.items()}, x.astype(np.float64))
        for (_, col, a, _bn), name in zip(convs, R._conv_names(9)):
            z = col @ g[name + "/weights"].astype(np.float64).reshape(col.shape[1], -1) + g[name + "/biases"]
            assert np.abs(z).min() > 0.05, name           # the premise of the tie-free construction
    assert np.abs(m.forward(x) - o.forward(x)).max() < 1e-4
    if gan:
        got = m.engine.d_backward(x[:, None], lab[:, None], None, train=True, apply=False).cpu().numpy()
        want, _ = o.d_tower(x, lab)
        assert np.allclose(got, want, rtol=1e-4), (got, want)
    got = m.engine.g_backward(x[:, None], lab[:, None], None, train=True, reuse=gan, apply=False).cpu().numpy()
    want, wg, _ = o.g_tower(x, lab)
    assert np.allclose(got, want, rtol=1e-4, atol=1e-7), (got, want)
    gr = split_flat(m.engine.get_grads(NET_G).cpu().numpy(), m.engine.tensor_table(NET_G))
    for k in wg:
        assert rel_err(gr[k].reshape(wg[k].shape), wg[k]) < 2e-3, k
    for i in range(3):          # trajectories: the north_star's 1e-3 (two frames through a clipped discriminator amplify fp32 rounding)
        if gan:
            assert np.allclose(np.ravel(m.d_step(x, lab)), o.d_step(x, lab), rtol=1e-3)
            assert np.allclose(np.ravel(m.g_step(x, lab, reuse_g_forward=True)), o.g_step(x, lab), rtol=1e-3)
        else:
            assert np.allclose(np.ravel(m.step(x, lab)), np.ravel(o.g_step(x, lab))[1:], rtol=1e-3)
    gv, _ = m.get_vars()
    for k in o.g:
        assert gv[k].shape == o.g[k].shape and rel_err(gv[k], o.g[k]) < 1e-3, k


def rced_gradient_norms():
    """(worker of tests/test_gpu_placement.py::test_rced_weight_gradient_geometries_agree) the reference's frame geometry (257 x 11,
    multi-strip: the geometry RSRGAN_WGRAD_DH acts on), 3 frames, random weights and biases: one supervised backward, the norm of
    every generator gradient"""
    from oracle import rced_oracle as R
    from rsrgan_amd.trainer import DNNTrainer
    N = 3
    cfg = R.RcedCfg(input_dim=257, output_dim=5, left_context=5, right_context=5, d_units=18, d_hidden=2, filters_num=R.FILTERS_NUM)
    rng = np.random.default_rng(78)
    g = {k: v.astype(np.float32) for k, v in R.init_params(R.g_param_specs(cfg), rng).items()}
    for k in g:
        if k.endswith("biases"):
            g[k] = rng.normal(0.05, 0.1, g[k].shape).astype(np.float32)
    d = {k: v.astype(np.float32) for k, v in DO.init_params(DO.d_param_specs(cfg), rng, relu_init=True).items()}
    args = SimpleNamespace(batch_size=N, input_dim=cfg.input_dim, output_dim=cfg.output_dim, left_context=5, right_context=5,
                           g_type="rced", keep_prob=1.0, batch_norm=False, num_gpu=1, save_dir=None, l2_scale=0.0,
                           g_learning_rate=1e-3, d_learning_rate=2e-3, init_mse_weight=10.0, disc_updates=1, gen_updates=1)
    m = DNNTrainer(None, args, ["gpu:0"], net_overrides=dict(g_layers=9, g_cells=32, d_layers=cfg.d_hidden, d_cells=cfg.d_units))
    m.set_vars(g, d)
    x = rng.standard_normal((N, cfg.fed_dim)).astype(np.float32); lab = rng.standard_normal((N, cfg.output_dim)).astype(np.float32)
    m.engine.g_backward(x[:, None], lab[:, None], None, train=True, reuse=False, apply=False)
    gg = split_flat(m.engine.get_grads(NET_G).cpu().numpy(), m.engine.tensor_table(NET_G))
    return {k: float(np.linalg.norm(v.astype(np.float64))) for k, v in gg.items()}


def test_rced_reference_frame_random_relu_masks():
    """The reference's frame (257-dim LPS x splice 11, filter table of models/rced.py:90-114: the 4 x 65-column strip path) with
    RANDOM biases and weights, so every channel's ReLU mask varies over the positions.  A frame has ~1 M ReLU units; a handful of
    pre-activations land within fp32 rounding of the kink, where fp32 and the fp64 oracle legitimately pick different
    subgradients.  The test therefore allows exactly that and nothing else: the HIP gradient must equal the oracle's plus a
    combination of the oracle's own linear responses to flipping the masks of the units with |z| < EPS -- a bounded number of
    them, each taken fully or not at all."""
    from oracle import rced_oracle as R
    from rsrgan_amd.trainer import DNNTrainer
    EPS, N = 2e-5, 1
    cfg = R.RcedCfg(input_dim=257, output_dim=5, left_context=5, right_context=5, d_units=18, d_hidden=2, filters_num=R.FILTERS_NUM)
    rng = np.random.default_rng(77)
    g = {k: v.astype(np.float32) for k, v in R.init_params(R.g_param_specs(cfg), rng).items()}
    for k in g:
        if k.endswith("biases"):
            g[k] = rng.normal(0.05, 0.1, g[k].shape).astype(np.float32)
    d = {k: v.astype(np.float32) for k, v in DO.init_params(DO.d_param_specs(cfg), rng, relu_init=True).items()}
    args = SimpleNamespace(batch_size=N, input_dim=cfg.input_dim, output_dim=cfg.output_dim, left_context=5, right_context=5,
                           g_type="rced", keep_prob=1.0, batch_norm=False, num_gpu=1, save_dir=None, l2_scale=0.0,
                           g_learning_rate=1e-3, d_learning_rate=2e-3, init_mse_weight=10.0, disc_updates=1, gen_updates=1)
    m = DNNTrainer(None, args, ["gpu:0"], net_overrides=dict(g_layers=9, g_cells=32, d_layers=cfg.d_hidden, d_cells=cfg.d_units))
    m.set_vars(g, d)
    x = rng.standard_normal((N, cfg.fed_dim)).astype(np.float32); lab = rng.standard_normal((N, cfg.output_dim)).astype(np.float32)
    P = {k: v.astype(np.float64) for k, v in g.items()}
    y, cache = R.rced_fwd(cfg, P, x.astype(np.float64))
    convs = cache[0]
    names = R._conv_names(9)
    assert np.abs(m.forward(x) - y).max() < 1e-4
    # ReLU masks do vary within a channel (what the tie-free W=257 cases of test_rced_generator_matches_oracle cannot cover)
    frac = [(c[2] > 0).mean(0) for c in convs]
    assert sum(int(((f > 0.05) & (f < 0.95)).sum()) for f in frac) > 100
    near = []
    for i, (_, col, a, _bn) in enumerate(convs):
        z = col @ P[names[i] + "/weights"].reshape(col.shape[1], -1) + P[names[i] + "/biases"]
        near += [(i, int(p), int(c)) for p, c in zip(*np.nonzero(np.abs(z) < EPS))]
    assert len(near) < 200, len(near)
    dy = 0.5 * cfg.output_dim * 2.0 * (y - lab) / y.size          # d (0.5 * Dout * mse) / dy   (dnn_trainer.py:139-141)
    trace = {}
    base = R.rced_bwd(cfg, P, cache, dy, trace=trace)
    got = m.engine.g_backward(x[:, None], lab[:, None], None, train=True, reuse=False, apply=False).cpu().numpy()
    assert np.isclose(got[1], 0.5 * cfg.output_dim * np.mean((y - lab) ** 2), rtol=1e-4)
    hip = {k: v.reshape(base[k].shape).astype(np.float64)
           for k, v in split_flat(m.engine.get_grads(NET_G).cpu().numpy(), m.engine.tensor_table(NET_G)).items()}
    keys = sorted(base)
    scale = {k: 1.0 / max(np.linalg.norm(base[k]), 1e-30) for k in keys}
    vec = lambda gr: np.concatenate([(gr[k] * scale[k]).ravel() if k in gr else np.zeros(base[k].size) for k in keys])
    cols = []
    for i, p, c in near:                   # flipping unit (i, p, c): the gradient w.r.t. its pre-activation changes by +-trace
        dp = np.zeros_like(trace[i])
        dp[p, c] = -trace[i][p, c] if convs[i][2][p, c] > 0 else trace[i][p, c]
        cols.append(vec(R.rced_bwd(cfg, P, cache, None, start=(i, dp))))
    resid = vec(hip) - vec(base)
    if cols:
        A = np.stack(cols, 1)
        live = np.linalg.norm(A, axis=0) > 1e-6                     # a unit nobody's gradient reaches decides nothing
        coef = np.linalg.lstsq(A[:, live], resid, rcond=None)[0]
        assert np.all(np.abs(coef - np.round(coef)) < 0.1) and np.all((coef > -0.1) & (coef < 1.1)), coef
        resid = resid - A[:, live] @ np.round(coef)
    off = 0
    for k in keys:                         # per tensor: what is left after the admissible flips is fp32 rounding
        n = base[k].size
        assert np.linalg.norm(resid[off:off + n]) < 2e-3, (k, np.linalg.norm(resid[off:off + n]), len(near))
        off += n


@pytest.mark.parametrize("N,gan,ctx,width", [(6, False, (2, 1), 9), (5, True, (2, 2), 21), (4, False, (1, 1), -11), (3, True, (1, 1), 70)])
def test_rced_batch_norm_matches_oracle(N, gan, ctx, width):
    """run_dnn.sh:129-134 (--g_type=rced --batch_norm=true): relu(batch_norm(conv2d, scale=True, renorm=True)) per output channel over
    [N, S, W] (models/rced.py:67-72,97-99), under DNNTrainer and paired with a batch-normalised discriminator_dnn, vs
    oracle/rced_oracle.py + oracle/bn_renorm.py: tower losses, every gradient, the update ops, Adam steps, the cross_validation twin."""
    from oracle import rced_oracle as R
    from rsrgan_amd import GAN
    from rsrgan_amd.trainer import DNNTrainer
    full = width < 0
    width = abs(width)
    cfg = R.RcedCfg(input_dim=width, output_dim=5, left_context=ctx[0], right_context=ctx[1], d_units=20, d_hidden=2, batch_norm=True,
                    filters_num=R.FILTERS_NUM if full else (4,) * 9)
    rng = np.random.default_rng(100 + N)
    g = R.init_params(R.g_param_specs(cfg), rng)
    d = {k: (2.0 * v if k.endswith("weights") else v) for k, v in DO.init_params(DO.d_param_specs(cfg), rng, relu_init=True).items()}
    for p in (g, d):                         # a state some way into training: r != 1, d != 0
        w = rng.uniform(0.2, 0.6)
        for k in p:
            if k.endswith("biases") or k.endswith("/beta"):
                p[k] = rng.normal(0.05, 0.1, p[k].shape)
            elif k.endswith("/gamma"):
                p[k] = rng.uniform(0.7, 1.3, p[k].shape)
            elif k.endswith("renorm_mean"):
                p[k] = w * rng.normal(0, 0.2, p[k].shape)
            elif k.endswith("renorm_stddev"):
                p[k] = w * rng.uniform(0.3, 1.0, p[k].shape)
            elif k.endswith("_weight"):
                p[k] = np.float64(w)
            elif k.endswith("moving_mean"):
                p[k] = rng.normal(0, 0.2, p[k].shape)
            elif k.endswith("moving_variance"):
                p[k] = rng.uniform(0.3, 1.0, p[k].shape)
    g = {k: np.asarray(v, np.float32) for k, v in g.items()}
    d = {k: np.asarray(v, np.float32) for k, v in d.items()}
    ov = dict(g_layers=9, g_cells=32 if full else 4, d_layers=cfg.d_hidden, d_cells=cfg.d_units)

    class RcedGan(GAN):
        G_TYPES = ("dnn", "rced")

    def build(cv, gp, dp):
        args = SimpleNamespace(batch_size=N, input_dim=cfg.input_dim, output_dim=cfg.output_dim, left_context=cfg.left_context,
                               right_context=cfg.right_context, g_type="rced", keep_prob=1.0, batch_norm=True, num_gpu=1, save_dir=None,
                               l2_scale=1e-3, g_learning_rate=1e-3, d_learning_rate=2e-3, init_mse_weight=10.0, disc_updates=1, gen_updates=1)
        if gan:
            m = RcedGan(None, args, ["gpu:0"], cross_validation=cv, net_overrides=ov)
            o = R.GanRcedOracle(cfg, gp, dp, l2_scale=1e-3, g_learning_rate=float(np.float32(1e-3)), d_learning_rate=float(np.float32(2e-3)),
                                cross_validation=cv)
        else:
            m = DNNTrainer(None, args, ["gpu:0"], cross_validation=cv, net_overrides=ov)
            o = R.GanRcedOracle(cfg, gp, dp, l2_scale=1e-3, g_learning_rate=float(np.float32(1e-3)), mse_lambda=1.0, cross_validation=cv)
            o.supervised = True
        m.set_vars({k: np.asarray(v, np.float32) for k, v in gp.items()}, {k: np.asarray(v, np.float32) for k, v in dp.items()})
        return m, o

    m, o = build(False, g, d)
    table = [(n, m._tf_shape(n, s)) for n, s, _ in m.engine.tensor_table(NET_G)]
    assert table == [(n, tuple(s)) for n, s in R.g_param_specs(cfg)], table

    def cmp_vars(mm, oo, tol=1e-3):
        gv, dv = mm.get_vars()
        for got, want in ((gv, oo.g), (dv, oo.d if gan else {})):
            for k in want:
                assert got[k].shape == np.shape(want[k]) and (rel_err(got[k], want[k]) < tol or np.abs(got[k] - want[k]).max() < 1e-6), k

    x = rng.standard_normal((N, cfg.fed_dim)).astype(np.float32); lab = rng.standard_normal((N, cfg.output_dim)).astype(np.float32)
    assert np.abs(m.forward(x) - o.forward(x)).max() < 2e-4
    if gan:
        got = m.engine.d_backward(x[:, None], lab[:, None], None, train=True, apply=False).cpu().numpy()
        want, wd = o.d_tower(x, lab)
        assert np.allclose(got, want, rtol=2e-4), (got, want)
        gr = split_flat(m.engine.get_grads(NET_D).cpu().numpy(), m.engine.tensor_table(NET_D))
        for k in wd:
            assert rel_err(gr[k], wd[k]) < 2e-3, k
        cmp_vars(m, o)
    got = m.engine.g_backward(x[:, None], lab[:, None], None, train=True, reuse=False, apply=False).cpu().numpy()
    want, wg, _ = o.g_tower(x, lab)
    assert np.allclose(got, want, rtol=2e-4, atol=1e-7), (got, want)
    gr = split_flat(m.engine.get_grads(NET_G).cpu().numpy(), m.engine.tensor_table(NET_G))
    for k in wg:
        assert rel_err(gr[k].reshape(wg[k].shape), wg[k]) < 3e-3, k
    cmp_vars(m, o)
    x2 = (1.3 * rng.standard_normal((N, cfg.fed_dim)) + 0.2).astype(np.float32)
    for xb in (x2, x):
        if gan:
            assert np.allclose(np.ravel(m.d_step(xb, lab)), o.d_step(xb, lab), rtol=1e-3)
            assert np.allclose(np.ravel(m.g_step(xb, lab, reuse_g_forward=True)), o.g_step(xb, lab), rtol=1e-3)
        else:
            assert np.allclose(np.ravel(m.step(xb, lab)), np.ravel(o.g_step(xb, lab))[1:], rtol=1e-3)
    cmp_vars(m, o)
    mcv, ocv = build(True, o.g, o.d)                        # is_training=False: moving statistics, no update ops
    if gan:
        assert np.allclose(np.ravel(mcv.g_step(x, lab, train=False)), ocv.g_step(x, lab, train=False), rtol=1e-3)
    else:
        assert np.allclose(np.ravel(mcv.step(x, lab, train=False)), np.ravel(ocv.g_step(x, lab, train=False))[1:], rtol=1e-3)
    assert np.abs(mcv.forward(x) - ocv.forward(x)).max() < 2e-4
    cmp_vars(mcv, ocv)


def test_rced_gan_full_size_properties():
    """BASELINE.json configs[3] at its full size (R-CED on 257-dim LPS +-5 frames + discriminator_dnn 297-4x1024-1, N = 6400 frames =
    B 64 x T 100): too big for the fp64 oracle, so size-independent properties: finite losses, tower-mean linearity (the gradient
    of a batch of 6400 frames is the mean of the gradients of its two halves of 3200; models/gan.py:158-175 means
    over frames), and permutation invariance of the frame order."""
    import torch
    from rsrgan_amd import GAN

    class RcedGan(GAN):
        G_TYPES = ("dnn", "rced")

    def model(n):
        args = SimpleNamespace(batch_size=n, input_dim=257, output_dim=40, left_context=5, right_context=5, g_type="rced", keep_prob=1.0,
                               batch_norm=False, num_gpu=1, save_dir=None, l2_scale=0.0, g_learning_rate=1e-3, d_learning_rate=1e-4,
                               init_mse_weight=10.0, disc_updates=1, gen_updates=1)
        return RcedGan(None, args, ["gpu:0"], seed=4321)

    N = 6400
    rng = np.random.default_rng(7)
    x = rng.standard_normal((N, 1, 257 * 11)).astype(np.float32); lab = rng.standard_normal((N, 1, 40)).astype(np.float32)
    big = model(N)
    dl = big.engine.d_backward(x, lab, None, train=True, apply=False).cpu().numpy()
    gd = big.engine.get_grads(1).cpu().numpy().copy()
    gl = big.engine.g_backward(x, lab, None, train=True, reuse=True, apply=False).cpu().numpy()
    gg = big.engine.get_grads(NET_G).cpu().numpy().copy()
    assert np.all(np.isfinite(dl)) and np.all(np.isfinite(gl)) and np.all(np.isfinite(gg)) and np.all(np.isfinite(gd))
    # permutation of the frames: same losses and gradients up to fp32 summation order
    perm = rng.permutation(N)
    dl_p = big.engine.d_backward(x[perm], lab[perm], None, train=True, apply=False).cpu().numpy()
    gl_p = big.engine.g_backward(x[perm], lab[perm], None, train=True, reuse=True, apply=False).cpu().numpy()
    gg_p = big.engine.get_grads(NET_G).cpu().numpy()
    assert np.allclose(dl, dl_p, rtol=1e-4) and np.allclose(gl, gl_p, rtol=1e-4), (dl, dl_p, gl, gl_p)
    assert rel_err(gg_p, gg) < 1e-3
    gv, dv = big.get_vars()
    del big
    torch.cuda.empty_cache()
    half = model(N // 2)
    half.set_vars(gv, dv)
    acc_l = np.zeros(4); acc_g = np.zeros_like(gg, dtype=np.float64); acc_dl = np.zeros(3)
    for k in range(2):
        sl = slice(k * N // 2, (k + 1) * N // 2)
        acc_dl += half.engine.d_backward(x[sl], lab[sl], None, train=True, apply=False).cpu().numpy() / 2
        acc_l += half.engine.g_backward(x[sl], lab[sl], None, train=True, reuse=True, apply=False).cpu().numpy() / 2
        acc_g += half.engine.get_grads(NET_G).cpu().numpy() / 2
    assert np.allclose(acc_dl, dl, rtol=1e-4) and np.allclose(acc_l, gl, rtol=1e-4), (acc_dl, dl, acc_l, gl)
    assert rel_err(acc_g, gg) < 1e-3


def test_frame_level_outer_loop_on_gpu(tmp_path):
    """scripts/train_gan_dnn.py end to end on the HIP path with --batch_norm=true: epochs of D-runs / G-runs on fresh random-shuffle
    batches, the cross_validation fetches, accept / reject with checkpoints, the decay rule, then decode -> feats.ark with the
    moving averages of the trainable variables and the batch-norm statistics."""
    from rsrgan_amd import run_gan_dnn as RD
    from rsrgan_amd.io import ArkReader, ArkWriter
    rng = np.random.default_rng(0)
    din, dout = 6, 4
    A = rng.standard_normal((din, dout)) * 0.5

    def data(tag, n):
        wi, wl = ArkWriter(str(tmp_path / (tag + "_in.scp"))), ArkWriter(str(tmp_path / (tag + "_lab.scp")))
        for i in range(n):
            T = int(rng.integers(60, 100))
            x = rng.standard_normal((T, din)) * 2 + 1
            wi.write_next_utt(str(tmp_path / (tag + "_in.ark")), "%s%02d" % (tag, i), x)
            wl.write_next_utt(str(tmp_path / (tag + "_lab.ark")), "%s%02d" % (tag, i), (x - 1) / 2 @ A - 1)
        wi.close(); wl.close()
        return str(tmp_path / (tag + "_in.scp")), str(tmp_path / (tag + "_lab.scp"))
    tr, cv, te = data("tr", 40), data("cv", 8), data("te", 3)
    np.savez(tmp_path / "train_cmvn.npz", mean_inputs=np.full(din, 1.0), stddev_inputs=np.full(din, 2.0),
             mean_labels=np.full(dout, -1.0), stddev_labels=np.ones(dout))
    FLAGS, _ = RD.build_parser().parse_known_args([
        "--data_dir", str(tmp_path), "--tr_inputs_scp", tr[0], "--tr_labels_scp", tr[1], "--cv_inputs_scp", cv[0], "--cv_labels_scp", cv[1],
        "--test_inputs_scp", te[0], "--input_dim", str(din), "--output_dim", str(dout), "--left_context", "1", "--right_context", "1",
        "--batch_size", "64", "--min_epoches", "2", "--max_epoches", "4", "--keep_lr", "1", "--save_dir", str(tmp_path / "exp"),
        "--g_learning_rate", "0.003", "--d_learning_rate", "0.001", "--batch_norm", "true", "--init_mse_weight", "10", "--num_threads", "2",
        "--gen_updates", "2"])
    ov = dict(g_layers=2, g_cells=32, d_layers=2, d_cells=16)
    logs = []
    hist = RD.train(FLAGS, log=logs.append, net_overrides=ov)
    text = "\n".join(logs)
    assert 1 <= len(hist) <= 4 and np.all(np.isfinite(hist))
    assert "CROSSVAL.LOSS PRERUN" in text and "Nnet Accepted" in text and "Training Done." in text
    assert os.path.exists(tmp_path / "exp" / "checkpoint") and os.path.exists(tmp_path / "batch_num.txt")
    first = float(text.split("CROSSVAL.LOSS PRERUN")[1].split("g_mse_loss = ")[1].split(",")[0])
    last = float(text.split("(CROSS AVG.LOSS)")[-1].split("g_mse_loss = ")[1].split(",")[0])
    assert last < 0.7 * first, (first, last)                 # the supervised term learns the linear map
    import glob
    from rsrgan_amd import summary as S                      # model.writer.add_summary once per epoch pass (train_gan_dnn.py:132-134,195-196)
    for sub in ("train", "eval"):
        ev = S.read_events(glob.glob(str(tmp_path / "exp" / sub / "events.out.tfevents.*"))[0])
        assert len(ev) >= 2 and ev[0][3] == "brain.Event:2" and np.isfinite(ev[1][2]["g_loss"]) and ev[1][2]["g_clean"]["num"] == 64 * dout
    FLAGS.decode = True
    scp = RD.decode(FLAGS, log=logs.append, net_overrides=ov)
    r, src = ArkReader(), ArkReader()
    r(scp); src(te[0])
    assert r.utt_ids == src.utt_ids
    for i in range(len(r.utt_ids)):
        out = r.read_utt_data_from_index(i)
        assert out.shape == (src.read_utt_data_from_index(i).shape[0], dout) and np.all(np.isfinite(out))
