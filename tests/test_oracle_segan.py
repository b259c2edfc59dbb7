"""What pins oracle/segan_oracle.py (the reference's TF 1.4 graph cannot run here): the TF SAME / conv2d_transpose index arithmetic
against direct loops, VBN and RMSProp closed forms, central differences through the whole D-run and G-run graphs (including the
path through the reference-batch statistics), and the variable tables at the reference's sizes."""
import numpy as np
import torch

from oracle import segan_oracle as S


def _loop_downconv(x, W, b):
    B, L, Cin = x.shape
    k, _, _, Cout = W.shape
    out, pl, _ = S.same_pad(L, k)
    y = np.zeros((B, out, Cout))
    for o in range(out):
        for dk in range(k):
            i = 2 * o + dk - pl
            if 0 <= i < L:
                y[:, o] += x[:, i] @ W[dk, 0]
    return y + b


def _loop_deconv(x, W, b, out_len):
    B, Lin, Cin = x.shape
    k, _, Cout, _ = W.shape
    lin, pl, _ = S.same_pad(out_len, k)
    assert lin == Lin
    y = np.zeros((B, out_len, Cout))
    for o in range(Lin):
        for dk in range(k):
            i = 2 * o + dk - pl
            if 0 <= i < out_len:
                y[:, i] += x[:, o] @ W[dk, 0].T                      # W[dk, 0] is [Cout, Cin]
    return y + b


def test_conv_index_arithmetic_against_direct_loops():
    rng = np.random.default_rng(0)
    for L, k, cin, cout in [(17, 20, 2, 3), (16, 20, 1, 4), (2827, 31, 1, 2), (9, 31, 3, 2), (1, 20, 2, 2), (2, 3, 2, 2), (45, 31, 2, 1)]:
        x = rng.standard_normal((2, L, cin)); W = rng.standard_normal((k, 1, cin, cout)); b = rng.standard_normal(cout)
        got = S.downconv(torch.tensor(x), torch.tensor(W), torch.tensor(b)).numpy()
        assert got.shape[1] == -(-L // 2)
        assert np.allclose(got, _loop_downconv(x, W, b), atol=1e-12), (L, k)
        lin = -(-L // 2)
        xd = rng.standard_normal((2, lin, cin)); Wd = rng.standard_normal((k, 1, cout, cin))
        got = S.deconv(torch.tensor(xd), torch.tensor(Wd), torch.tensor(b), L).numpy()
        assert np.allclose(got, _loop_deconv(xd, Wd, b, L), atol=1e-12), (L, k)
        # conv2d_transpose is the adjoint of the strided convolution with the same filter tensor
        Wc = rng.standard_normal((k, 1, cin, cout))
        u = rng.standard_normal((2, L, cin)); v = rng.standard_normal((2, lin, cout))
        lhs = (S.downconv(torch.tensor(u), torch.tensor(Wc), None).numpy() * v).sum()
        rhs = (u * S.deconv(torch.tensor(v), torch.tensor(Wc), None, L).numpy()).sum()
        assert np.isclose(lhs, rhs, rtol=1e-10)


def test_vbn_and_rmsprop_known_answers():
    rng = np.random.default_rng(1)
    h = torch.tensor(rng.standard_normal((4, 7, 3)) * 3 + 2)
    m, q = S.vbn_stats(h)
    y = S.vbn_apply(h, m, q, torch.ones(3), torch.zeros(3), 0.0)
    assert np.allclose(y.mean(dim=(0, 1)).numpy(), 0, atol=1e-12) and np.allclose((y * y).mean(dim=(0, 1)).numpy(), 1, atol=1e-12)
    # live pass on the reference batch itself: the mixed statistics are the reference statistics
    c = 1.0 / 5.0
    assert np.allclose((c * m + (1 - c) * m).numpy(), m.numpy())
    cfg = S.SeganCfg(input_len=12, output_dim=3, g_depths=(2, 3), d_depths=(2, 2))
    g = S.init_params(S.g_param_specs(cfg), rng); d = S.init_params(S.d_param_specs(cfg), rng)
    o = S.SeganOracle(cfg, g, d, batch_size=2, d_learning_rate=0.5)
    grads = {k: rng.standard_normal(v.shape) for k, v in o.d.items()}
    before = {k: v.copy() for k, v in o.d.items()}
    S.SeganOracle._rmsprop(o.d, o.d_ms, grads, 0.5)
    for k in grads:                                                  # rms slot starts at ONE (TF 1.4), epsilon inside the root
        ms = 0.9 + 0.1 * grads[k] ** 2
        assert np.allclose(o.d_ms[k], ms) and np.allclose(o.d[k], before[k] - 0.5 * grads[k] / np.sqrt(ms + 1e-10))


def test_variable_tables_at_the_reference_sizes():
    cfg = S.SeganCfg()
    gs, ds = dict(S.g_param_specs(cfg)), dict(S.d_param_specs(cfg))
    assert gs["g_ae/enc_0/W"] == (20, 1, 1, 16) and gs["g_ae/enc_10/W"] == (20, 1, 512, 1024)
    assert gs["g_ae/dec_0/W"] == (20, 1, 512, 2048) and gs["g_ae/dec_1/W"] == (20, 1, 256, 1024) and gs["g_ae/dec_10/W"] == (20, 1, 1, 32)
    assert gs["g_ae/dense/kernel"] == (2827, 40) and "g_ae/dec_prelu_10/alpha" not in gs and gs["g_ae/dec_prelu_9/alpha"] == (16,)
    assert S.enc_lengths(2827, 11) == [2827, 1414, 707, 354, 177, 89, 45, 23, 12, 6, 3, 2]
    assert ds["d_model/d_block_0/downconv/W"] == (31, 1, 1, 16) and ds["d_model/logits_conv/W"] == (31, 1024, 1)
    assert ds["d_model/fully_connected/weights"] == (2, 1)          # 2867 joint samples through 11 stride-2 blocks
    assert len(ds) == 11 * 4 + 3 and len(gs) == 11 * 3 + 10 * 3 + 2 + 2


def test_gradients_by_central_differences():
    rng = np.random.default_rng(2)
    cfg = S.SeganCfg(input_len=13, output_dim=3, g_depths=(2, 3), d_depths=(2, 3), g_kwidth=4, d_kwidth=5)
    g = S.init_params(S.g_param_specs(cfg), rng); d = S.init_params(S.d_param_specs(cfg), rng)
    for p in (g, d):
        for k in p:
            p[k] = p[k] + 0.3 * rng.standard_normal(p[k].shape)     # nothing at its special initial value
    B = 3
    x = rng.standard_normal((B, 13)); lab = rng.standard_normal((B, 3)); z = rng.standard_normal((B, 4, 3))
    nz = [0.3 * rng.standard_normal((B, 16)) for _ in range(3)]
    o = S.SeganOracle(cfg, g, d, batch_size=B, l1_lambda=7.0)
    ld, gd = o.d_tower(x, lab, z, *nz)
    lg, gg, Gx = o.g_tower(x, lab, z, nz[0], nz[2])
    assert np.isclose(ld[2], ld[0] + ld[1]) and np.isclose(lg[2], lg[0] + lg[1]) and Gx.shape == (B, 3)
    assert np.isclose(lg[1], 7.0 * np.abs(Gx - lab).mean())
    eps = 1e-6
    def fd(P, name, idx, fn):
        old = P[name][idx]
        P[name][idx] = old + eps; up = fn()
        P[name][idx] = old - eps; dn = fn()
        P[name][idx] = old
        return (up - dn) / (2 * eps)
    for name in ("d_model/d_block_0/downconv/W", "d_model/d_block_0/d_vbn_0/gamma", "d_model/d_block_1/downconv/b", "d_model/logits_conv/W",
                 "d_model/fully_connected/weights"):
        idx = tuple(int(rng.integers(0, s)) for s in o.d[name].shape)
        num = fd(o.d, name, idx, lambda: o.d_tower(x, lab, z, *nz)[0][2])
        assert np.isclose(num, gd[name][idx], rtol=1e-5, atol=1e-8), name
    for name in ("g_ae/enc_0/W", "g_ae/enc_prelu_1/alpha", "g_ae/dec_0/W", "g_ae/dec_1/b", "g_ae/dense/kernel"):
        idx = tuple(int(rng.integers(0, s)) for s in o.g[name].shape)
        num = fd(o.g, name, idx, lambda: o.g_tower(x, lab, z, nz[0], nz[2])[0][2])
        assert np.isclose(num, gg[name][idx], rtol=1e-5, atol=1e-8), name
    # the reference ("dummy") pass carries gradient: with its noise changed the D gradients change, the generator's do not need it
    _, gd2 = o.d_tower(x, lab, z, nz[0] * 0.0, nz[1], nz[2])
    assert not np.allclose(gd2["d_model/d_block_0/downconv/W"], gd["d_model/d_block_0/downconv/W"])
    # three RMSProp steps run and move both nets
    d0 = o.d["d_model/logits_conv/W"].copy(); g0 = o.g["g_ae/dense/kernel"].copy()
    for _ in range(3):
        o.d_step(x, lab, z, *nz); o.g_step(x, lab, z, nz[0], nz[2])
    assert not np.allclose(d0, o.d["d_model/logits_conv/W"]) and not np.allclose(g0, o.g["g_ae/dense/kernel"])
