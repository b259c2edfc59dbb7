"""Pins the R-CED oracle (oracle/rced_oracle.py): SAME convolution against a direct loop, im2col/col2im adjointness,
central differences of the trainer loss, parameter table."""
import numpy as np

from oracle import rced_oracle as R


def _small():
    return R.RcedCfg(input_dim=7, output_dim=3, left_context=1, right_context=1, filters_num=(3, 4, 2), filters_width=(5, 3, 3))


def test_param_table_of_the_reference_network():
    cfg = R.RcedCfg(input_dim=40, output_dim=40)          # run_dnn.sh:129-140 (splice 11)
    specs = R.g_param_specs(cfg)
    assert specs[0] == ("g_model/Conv/weights", (11, 13, 1, 12)) and specs[-2][1] == (11 * 40 * 12, 40)
    assert [s[0] for s in specs[2:6]] == ["g_model/Conv_1/weights", "g_model/Conv_1/biases", "g_model/Conv_2/weights", "g_model/Conv_2/biases"]
    assert len(specs) == 20
    p = R.init_params(specs, np.random.default_rng(0))
    assert np.all(p["g_model/fully_connected/biases"] == 0.1) and np.all(p["g_model/Conv_3/biases"] == 0.0)


def test_same_convolution_matches_direct_loop():
    rng = np.random.default_rng(1)
    N, S, W, C, Co, kw = 2, 3, 6, 2, 3, 5
    x = rng.standard_normal((N, S, W, C)); F = rng.standard_normal((S, kw, C, Co))
    got = (R.im2col(x, S, kw) @ F.reshape(-1, Co)).reshape(N, S, W, Co)
    want = np.zeros_like(got)
    pt, pl = (S - 1) // 2, (kw - 1) // 2
    for n in range(N):
        for h in range(S):
            for w in range(W):
                for dh in range(S):
                    for dw in range(kw):
                        hh, ww = h + dh - pt, w + dw - pl
                        if 0 <= hh < S and 0 <= ww < W:
                            want[n, h, w] += x[n, hh, ww] @ F[dh, dw]
    assert np.allclose(got, want)
    # col2im is the adjoint of im2col: <im2col(x), c> == <x, col2im(c)>
    c = rng.standard_normal((N * S * W, S * kw * C))
    assert abs(np.sum(R.im2col(x, S, kw) * c) - np.sum(x * R.col2im(c, x.shape, S, kw))) < 1e-9


def test_trainer_loss_gradient_by_central_differences():
    cfg = _small()
    rng = np.random.default_rng(2)
    g = R.init_params(R.g_param_specs(cfg), rng)
    for k in g:
        if k.endswith("biases"):
            g[k] = rng.normal(0, 0.1, g[k].shape)
    o = R.GanRcedOracle(cfg, g, {}, mse_lambda=1.0, l2_scale=1e-2)
    o.supervised = True
    x = rng.standard_normal((4, cfg.fed_dim)); lab = rng.standard_normal((4, cfg.output_dim))
    losses, grads, y = o.g_tower(x, lab)
    assert y.shape == (4, 3) and losses[0] == 0.0
    for name in o.g:
        idx = tuple(rng.integers(0, s) for s in o.g[name].shape)
        old = o.g[name][idx]
        o.g[name][idx] = old + 1e-6; lp = o.g_tower(x, lab, want_grads=False)[0][3]
        o.g[name][idx] = old - 1e-6; lm = o.g_tower(x, lab, want_grads=False)[0][3]
        o.g[name][idx] = old
        fd = (lp - lm) / 2e-6
        assert abs(fd - grads[name][idx]) < 1e-6 * max(1.0, abs(fd)), (name, fd, grads[name][idx])
