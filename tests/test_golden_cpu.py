"""The oracle (fp64 numpy), its fp32 run and the torch-autograd twin against the committed golden
vectors (tests/golden/*.npz, produced by tests/golden/make_golden.py)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import rsrgan_oracle as O
from oracle import torch_twin as TT
from tests.helpers import rand_batch, rand_params, small_cfg

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_small(path):
    z = np.load(path)
    cfg = small_cfg(str(z["g_type"]))
    g = {k[3:]: z[k] for k in z.files if k.startswith("g0/")}
    d = {k[3:]: z[k] for k in z.files if k.startswith("d0/")}
    return z, cfg, g, d


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "small_*.npz"))))
def test_oracle_reproduces_small_fixtures(path):
    z, cfg, g, d = load_small(path)
    kw = dict(l2_scale=1e-3, g_learning_rate=float(np.float32(1e-3)), d_learning_rate=float(np.float32(5e-2)))
    x, lab, ln = z["x"], z["lab"], z["ln"]
    nr, nf = z["noise_real"].astype(np.float64), z["noise_fake"].astype(np.float64)
    for dtype, tol in ((np.float64, 1e-7), (np.float32, 2e-4)):     # fixtures store fp32 noise: 1e-7
        o = O.GanRnnOracle(cfg, g, d, batch_size=x.shape[0], dtype=dtype, **kw)
        assert np.allclose(o.forward(x, ln), z["y0"], rtol=tol, atol=tol)
        assert np.allclose(np.ravel(o.d_step(x, lab, ln, nr, nf)), z["d_step"], rtol=tol)
        assert np.allclose(np.ravel(o.g_step(x, lab, ln, nf)), z["g_step1"], rtol=tol)
        assert np.allclose(np.ravel(o.g_step(x, lab, ln, nf)), z["g_step2"], rtol=tol)
        for k in o.g:
            assert np.allclose(o.g[k], z["g1/" + k], rtol=50 * tol, atol=tol), k
    tw = TT.GanRnnTorchTwin(cfg, g, d, dtype=torch.float64, **kw)
    assert np.allclose(tw.d_step(x, lab, ln, nr, nf), z["d_step"], rtol=1e-7)
    assert np.allclose(tw.g_step(x, lab, ln, nf), z["g_step1"], rtol=1e-7)


@pytest.mark.parametrize("tag,cfg", [("lstm", O.NetCfg()), ("res_lstm_l", O.NetCfg.res_lstm_l())])
def test_twin_fp32_reproduces_reference_true_fixture(tag, cfg):
    """fp32 torch twin (the timed CPU baseline) on the reference's hard-coded sizes vs the fp64 fixture."""
    z = np.load(os.path.join(GOLD, "reftrue_%s.npz" % tag))
    g, d = rand_params(cfg, int(z["seed"]))
    x, lab, ln = rand_batch(cfg, int(z["B"]), int(z["T"]), int(z["seed"]) + 1, True)
    assert np.array_equal(ln, z["ln"])
    tw = TT.GanRnnTorchTwin(cfg, g, d, dtype=torch.float32)
    assert np.allclose(tw.d_step(x, lab, ln), z["d_step"], rtol=1e-4)
    assert np.allclose(tw.g_step(x, lab, ln), z["g_step1"], rtol=1e-4)
    assert np.allclose(tw.g_step(x, lab, ln), z["g_step2"], rtol=1e-4)
