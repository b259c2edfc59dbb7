"""BASELINE.json configs[4]: the SEGAN-style conv G/D (models/segan.py + generator.py:AEGenerator + discriminator.py + utils/bnorm.py)
on the HIP path (rsrgan_segan_* of include/rsrgan.h: window-view GEMMs for the strided convolutions, parity-class GEMMs for the
transposed ones) against the fp64 oracle (oracle/segan_oracle.py): generator output, tower losses, every gradient tensor of both
runs -- including the path through the reference-batch statistics of the virtual batch norm -- and three RMSProp steps."""
from types import SimpleNamespace

import numpy as np
import pytest

from oracle import segan_oracle as S
from tests.helpers import NET_D, NET_G, rel_err

pytestmark = pytest.mark.gpu


def _close(a, b, rtol=2e-3, scale=None):
    """relative L2.  The conv biases in front of a virtual batch norm have an exactly ZERO gradient (the normaliser removes any
    per-channel shift): there fp32 leaves the rounding of a sum of ~1e5 terms and a relative error means nothing, so such a
    tensor must vanish against `scale`, the gradient of its block's filter."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    if scale is not None and np.linalg.norm(b) <= 1e-9 * np.linalg.norm(scale):
        return np.linalg.norm(a) <= 1e-4 * np.linalg.norm(scale)
    return np.linalg.norm(a - b) <= rtol * np.linalg.norm(b) + 2e-6 * np.sqrt(b.size)


def _d_close(gd, wg, k):
    return _close(gd[k], wg[k], scale=wg[k[:-1] + "W"] if k.endswith("downconv/b") else None)


def _pair(B, L, U, depths, gk, dk, seed, g_nl="prelu", l1=100.0, g_lr=2e-4, d_lr=3e-4):
    from rsrgan_amd import SEGAN
    cfg = S.SeganCfg(input_len=L, output_dim=U, g_depths=tuple(depths), d_depths=tuple(depths), g_kwidth=gk, d_kwidth=dk, g_nl=g_nl)
    rng = np.random.default_rng(seed)
    g = S.init_params(S.g_param_specs(cfg), rng); d = S.init_params(S.d_param_specs(cfg), rng)
    for p in (g, d):                                       # nothing at its special initial value (alpha 0, biases 0, beta 0)
        for k in p:
            if not (k.endswith("/W") or k.endswith("kernel") or k.endswith("weights")):
                p[k] = p[k] + 0.1 * rng.standard_normal(p[k].shape)
            p[k] = p[k].astype(np.float32)
    args = SimpleNamespace(batch_size=B, input_dim=L, output_dim=U, left_context=0, right_context=0, g_type="ae", deconv_type="deconv",
                           bias_downconv=True, bias_deconv=True, bias_D_conv=True, g_nl=g_nl, init_noise_std=0.0, init_l1_weight=l1,
                           g_learning_rate=g_lr, d_learning_rate=d_lr, save_dir=None)
    m = SEGAN(None, args, ["gpu:0"], depths=tuple(depths), g_kwidth=gk, d_kwidth=dk)
    want_g = [(n, tuple(s)) for n, s in S.g_param_specs(cfg)]
    want_d = [(n, tuple(s)) for n, s in S.d_param_specs(cfg)]
    assert [(n, m._tf_shape(n, s)) for n, s, _ in m.tensor_table(NET_G)] == want_g
    assert [(n, m._tf_shape(n, s)) for n, s, _ in m.tensor_table(NET_D)] == want_d
    m.set_vars(g, d)
    o = S.SeganOracle(cfg, g, d, batch_size=B, g_learning_rate=float(np.float32(g_lr)), d_learning_rate=float(np.float32(d_lr)), l1_lambda=l1)
    return cfg, m, o, rng


def _batch(cfg, B, rng, noise=0.3):
    n = len(cfg.g_depths)
    x = rng.standard_normal((B, cfg.input_len)).astype(np.float32); lab = rng.standard_normal((B, cfg.output_dim)).astype(np.float32)
    z = rng.standard_normal((B, S.enc_lengths(cfg.input_len, n)[-1], cfg.g_depths[-1])).astype(np.float32)
    nz = [(noise * rng.standard_normal((B, cfg.input_len + cfg.output_dim))).astype(np.float32) for _ in range(3)]
    return x, lab, z, nz


@pytest.mark.parametrize("B,L,U,depths,gk,dk,g_nl", [
    (3, 37, 5, (16, 32, 16), 20, 31, "prelu"),             # odd lengths everywhere: 37 -> 19 -> 10 -> 5 ; joint 42 -> 21 -> 11 -> 6
    (2, 64, 8, (16, 16, 32, 32), 20, 31, "leakyrelu"),     # even lengths, the leaky-ReLU generator
    (4, 50, 4, (16, 32), 6, 5, "prelu"),                   # short filters (both parities 3 / 3 and 3 / 2 taps)
    (2, 300, 40, (16, 32, 32, 64, 64, 128), 20, 31, "prelu"),
])
def test_segan_runs_match_oracle(B, L, U, depths, gk, dk, g_nl):
    cfg, m, o, rng = _pair(B, L, U, depths, gk, dk, seed=B * 100 + L, g_nl=g_nl, l1=7.0)
    x, lab, z, nz = _batch(cfg, B, rng)
    y = m.forward(x, z); y_ref = o.forward(x, z)
    assert np.abs(y - y_ref).max() < 1e-4 * max(1.0, np.abs(y_ref).max()), np.abs(y - y_ref).max()
    got = m.d_step(x, lab, z, nz, apply=False); want, wg = o.d_tower(x, lab, z, *nz)
    assert np.allclose(got, want, rtol=1e-4), (got, want)
    gd = m.get_grads(NET_D)
    for k in wg:
        assert _d_close(gd, wg, k), ("D", k, rel_err(gd[k], wg[k]))
    got = m.g_step(x, lab, z, (nz[0], nz[2]), apply=False); want, wg, _ = o.g_tower(x, lab, z, nz[0], nz[2])
    assert np.allclose(got, want, rtol=1e-4), (got, want)
    gg = m.get_grads(NET_G)
    for k in wg:
        assert _close(gg[k], wg[k]), ("G", k, rel_err(gg[k], wg[k]))
    # the eval fetches change nothing and agree
    assert np.allclose(m.d_step(x, lab, z, nz, train=False), want if False else o.d_tower(x, lab, z, *nz)[0], rtol=1e-4)
    for _ in range(3):                                     # three RMSProp steps of both nets
        a = m.d_step(x, lab, z, nz); b = o.d_step(x, lab, z, *nz)
        assert np.allclose(a, b, rtol=1e-3), (a, b)
        a = m.g_step(x, lab, z, (nz[0], nz[2])); b = o.g_step(x, lab, z, nz[0], nz[2])
        assert np.allclose(a, b, rtol=1e-3), (a, b)
    gv, dv = m.get_vars()
    for k in o.g:
        assert gv[k].shape == o.g[k].shape and rel_err(gv[k], o.g[k]) < 1e-3, k
    for k in o.d:
        assert rel_err(dv[k], o.d[k]) < 1e-3, k
    gm, dm = m.get_vars(1)                                 # the rms slots started at one
    for k in o.d_ms:
        assert rel_err(dm[k], o.d_ms[k]) < 1e-3, k


def test_segan_waveform_chunk_against_oracle():
    """BASELINE.json configs[4]'s own geometry: 16384-sample chunks through the reference's 11 layers (16..1024 feature maps,
    kwidth 20 / 31), batch 2 (the oracle's fp64 autograd takes ~20 s here)."""
    B, L, U = 2, 16384, 40
    cfg, m, o, rng = _pair(B, L, U, S.DEPTHS, 20, 31, seed=5, l1=100.0)
    x, lab, z, nz = _batch(cfg, B, rng, noise=0.1)
    y = m.forward(x, z); y_ref = o.forward(x, z)
    assert np.abs(y - y_ref).mean() / np.abs(y_ref).mean() < 1e-3
    got = m.d_step(x, lab, z, nz, apply=False); want, wg = o.d_tower(x, lab, z, *nz)
    assert np.allclose(got, want, rtol=1e-4), (got, want)
    gd = m.get_grads(NET_D)
    for k in wg:
        assert _d_close(gd, wg, k), ("D", k, rel_err(gd[k], wg[k]))
    got = m.g_step(x, lab, z, (nz[0], nz[2]), apply=False); want, wg, _ = o.g_tower(x, lab, z, nz[0], nz[2])
    assert np.allclose(got, want, rtol=1e-4), (got, want)
    gg = m.get_grads(NET_G)
    for k in wg:
        assert _close(gg[k], wg[k]), ("G", k, rel_err(gg[k], wg[k]))
    for _ in range(3):
        a = m.d_step(x, lab, z, nz); b = o.d_step(x, lab, z, *nz)
        assert np.allclose(a, b, rtol=1e-3), (a, b)
        a = m.g_step(x, lab, z, (nz[0], nz[2])); b = o.g_step(x, lab, z, nz[0], nz[2])
        assert np.allclose(a, b, rtol=1e-3), (a, b)


def test_segan_at_the_benchmarked_batch_against_golden():
    """BASELINE.json configs[4] at B = 32 (what bench.py --net segan --batch 32 times): the virtual batch norm's 1 / (B + 1) mix
    (utils/bnorm.py:36-48) and every planner branch (stream-K cuts at M = 32 L, the fix-up pieces) differ from the B = 2 case above.
    The fp64 oracle ran in the build container (tests/golden/make_segan_golden.py: same seeds, same draws as _pair / _batch here) and
    left the towers' losses, a sample of G(x), the norm of every gradient tensor and -- after one RMSProp step of each net -- the
    next losses and the norm of every variable's change."""
    import os
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "segan_b32_l16384.npz"))
    B, L, U = int(gold["B"]), int(gold["L"]), int(gold["U"])
    assert (B, L, U) == (32, 16384, 40)
    cfg, m, o, rng = _pair(B, L, U, S.DEPTHS, 20, 31, seed=int(gold["seed"]), l1=100.0)
    x, lab, z, nz = _batch(cfg, B, rng, noise=0.1)
    g0, d0 = m.get_vars()
    y = m.forward(x, z)
    assert abs(np.abs(y).mean() - float(gold["y_abs_mean"])) < 1e-3 * float(gold["y_abs_mean"])
    assert np.abs(y[:, ::4] - gold["y_sample"]).mean() / np.abs(gold["y_sample"]).mean() < 1e-3
    got = m.d_step(x, lab, z, nz, apply=False)
    assert np.allclose(got, gold["d_losses"], rtol=1e-4), (got, gold["d_losses"])
    gd = m.get_grads(NET_D)
    for k in gd:
        want = float(gold["dgrad_norm/" + k])
        scale = float(gold["dgrad_norm/" + k[:-1] + "W"]) if k.endswith("downconv/b") else None
        if scale is not None and want <= 1e-9 * scale:                 # a bias in front of a VBN: exactly zero gradient (see _close)
            assert np.linalg.norm(gd[k]) <= 1e-4 * scale, ("D", k)
        else:
            assert abs(np.linalg.norm(gd[k].astype(np.float64)) - want) <= 2e-3 * want + 1e-9, ("D", k, np.linalg.norm(gd[k]), want)
    got = m.g_step(x, lab, z, (nz[0], nz[2]), apply=False)
    assert np.allclose(got, gold["g_losses"], rtol=1e-4), (got, gold["g_losses"])
    gg = m.get_grads(NET_G)
    for k in gg:
        want = float(gold["ggrad_norm/" + k])
        assert abs(np.linalg.norm(gg[k].astype(np.float64)) - want) <= 2e-3 * want + 1e-9, ("G", k, np.linalg.norm(gg[k]), want)
    a = m.d_step(x, lab, z, nz)
    assert np.allclose(a, gold["d_step"], rtol=1e-3), (a, gold["d_step"])
    a = m.g_step(x, lab, z, (nz[0], nz[2]))
    assert np.allclose(a, gold["g_step"], rtol=1e-3), (a, gold["g_step"])
    # (an fp32 variable moves in steps of its own ulp: where the RMSProp step is of that size -- the big decoder filters: |delta| ~ 1e-9
    # per element -- the norm of the change carries rounding noise of ~1e-7 of the VARIABLE's norm)
    g1, d1 = m.get_vars()
    for k in d1:
        want = float(gold["d1_delta_norm/" + k])
        assert abs(np.linalg.norm(d1[k].astype(np.float64) - d0[k]) - want) <= 5e-3 * want + 1e-7 * np.linalg.norm(d0[k]) + 1e-7, ("dD", k)
    for k in g1:
        want = float(gold["g1_delta_norm/" + k])
        assert abs(np.linalg.norm(g1[k].astype(np.float64) - g0[k]) - want) <= 5e-3 * want + 1e-7 * np.linalg.norm(g0[k]) + 1e-7, ("dG", k)
    a = m.d_step(x, lab, z, nz, apply=False)
    assert np.allclose(a, gold["d_next"], rtol=1e-3), (a, gold["d_next"])


def test_segan_checkpoint_and_device_draws(tmp_path):
    cfg, m, o, rng = _pair(2, 40, 4, (16, 16), 6, 5, seed=9)
    x, lab, z, nz = _batch(cfg, 2, rng)
    m.set_scalar("disc_noise_std", 0.2)
    a = m.d_step(x, lab); b = m.g_step(x, lab)             # z and the noise draws made on the device
    assert np.all(np.isfinite(a)) and np.all(np.isfinite(b))
    m.save(str(tmp_path), 3)
    g0, d0 = m.get_vars(); r0 = m.get_vars(1)
    m.d_step(x, lab); m.g_step(x, lab)
    assert not np.allclose(m.get_vars()[0]["g_ae/dense/kernel"], g0["g_ae/dense/kernel"])
    assert m.load(str(tmp_path))
    g1, d1 = m.get_vars(); r1 = m.get_vars(1)
    for k in g0:
        assert np.array_equal(g0[k], g1[k]) and np.array_equal(r0[0][k], r1[0][k])
    for k in d0:
        assert np.array_equal(d0[k], d1[k])
