"""HIP path (through the C ABI) against the committed golden vectors."""
import glob
import os

import numpy as np
import pytest

from oracle import rsrgan_oracle as O
from tests.helpers import NET_D, NET_G, args_for, overrides, rand_batch, rand_params, rel_err, small_cfg, split_flat

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _model(cfg, B, T, g, d, flags, **kw):
    from rsrgan_amd import GAN_RNN
    m = GAN_RNN(None, args_for(cfg, B, **kw), ["gpu:0"], max_frames=T, net_overrides=dict(overrides(cfg), flags=flags))
    m.set_vars(g, d)
    return m


@pytest.mark.parametrize("flags", [0, 1])
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "small_*.npz"))))
def test_small_fixtures(path, flags):
    z = np.load(path)
    cfg = small_cfg(str(z["g_type"]))
    g = {k[3:]: z[k] for k in z.files if k.startswith("g0/")}
    d = {k[3:]: z[k] for k in z.files if k.startswith("d0/")}
    x, lab, ln, nr, nf = z["x"], z["lab"], z["ln"], z["noise_real"], z["noise_fake"]
    B, T = x.shape[:2]
    m = _model(cfg, B, T, g, d, flags, l2_scale=1e-3, g_learning_rate=1e-3, d_learning_rate=5e-2)
    y = m.forward(x, ln)
    assert np.abs(y - z["y0"]).mean() / np.abs(z["y0"]).mean() < 1e-4
    got = m.engine.d_backward(x, lab, ln, nr, nf, train=True, apply=False).cpu().numpy()
    assert np.allclose(got, z["d_losses"], rtol=1e-4)
    gr = split_flat(m.engine.get_grads(NET_D).cpu().numpy(), m.engine.tensor_table(NET_D))
    for k in gr:
        assert rel_err(gr[k], z["dgrad/" + k]) < 2e-3, k
    got = m.engine.g_backward(x, lab, ln, nf, train=True, reuse=False, apply=False).cpu().numpy()
    assert np.allclose(got, z["g_losses"], rtol=1e-4)
    gr = split_flat(m.engine.get_grads(NET_G).cpu().numpy(), m.engine.tensor_table(NET_G))
    for k in gr:
        assert rel_err(gr[k], z["ggrad/" + k]) < 2e-3, k
    assert np.allclose(np.ravel(m.d_step(x, lab, ln, nr, nf)), z["d_step"], rtol=1e-4)
    assert np.allclose(np.ravel(m.g_step(x, lab, ln, nf, reuse_g_forward=True)), z["g_step1"], rtol=1e-4)
    assert np.allclose(np.ravel(m.g_step(x, lab, ln, nf)), z["g_step2"], rtol=1e-4)
    gv, dv = m.get_vars()
    for k in gv:
        assert rel_err(gv[k], z["g1/" + k]) < 1e-4, k
    for k in dv:
        assert rel_err(dv[k], z["d1/" + k]) < 1e-4, k


@pytest.mark.parametrize("flags", [0, 1])
@pytest.mark.parametrize("tag,cfg", [("lstm", O.NetCfg()), ("res_lstm_l", O.NetCfg.res_lstm_l())])
def test_reference_true_fixture(tag, cfg, flags):
    z = np.load(os.path.join(GOLD, "reftrue_%s.npz" % tag))
    g, d = rand_params(cfg, int(z["seed"]))
    B, T = int(z["B"]), int(z["T"])
    x, lab, ln = rand_batch(cfg, B, T, int(z["seed"]) + 1, True)
    m = _model(cfg, B, T, g, d, flags)
    y = m.forward(x, ln)
    assert np.allclose(y[:, :, ::8], z["y0_sample"], atol=2e-5)
    assert abs(np.abs(y).mean() / float(z["y0_abs_mean"]) - 1) < 1e-4
    got = m.engine.d_backward(x, lab, ln, None, None, train=True, apply=False).cpu().numpy()
    assert np.allclose(got, z["d_losses"], rtol=1e-4)
    gr = split_flat(m.engine.get_grads(NET_D).cpu().numpy(), m.engine.tensor_table(NET_D))
    for k in gr:
        assert abs(np.linalg.norm(gr[k]) / float(z["dgrad_norm/" + k]) - 1) < 2e-3, k
    got = m.engine.g_backward(x, lab, ln, None, train=True, reuse=True, apply=False).cpu().numpy()
    assert np.allclose(got, z["g_losses"], rtol=1e-4)
    gr = split_flat(m.engine.get_grads(NET_G).cpu().numpy(), m.engine.tensor_table(NET_G))
    for k in gr:
        assert abs(np.linalg.norm(gr[k]) / float(z["ggrad_norm/" + k]) - 1) < 2e-3, k
    assert np.allclose(np.ravel(m.d_step(x, lab, ln)), z["d_step"], rtol=1e-4)
    assert np.allclose(np.ravel(m.g_step(x, lab, ln, reuse_g_forward=True)), z["g_step1"], rtol=1e-4)
    assert np.allclose(np.ravel(m.g_step(x, lab, ln)), z["g_step2"], rtol=1e-4)
