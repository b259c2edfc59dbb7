"""Host-side mirror of the reference surface (GAN_RNN, train_one_iteration, exponential_decay,
save/load) exercised on CPU through the oracle-backed engine."""
import math
import queue

import numpy as np
import pytest

from oracle import rsrgan_oracle as O
from rsrgan_amd import GAN_RNN, eval_one_iteration, exponential_decay, train_one_iteration
from tests.helpers import OracleEngine, args_for, rand_batch, rand_params, small_cfg


def _model(cfg, B, seed=0, **kw):
    g, d = rand_params(cfg, seed)
    eng = OracleEngine(cfg, g, d, B, l2_scale=kw.get("l2_scale", 0.0))
    return GAN_RNN(None, args_for(cfg, B, **kw), ["cpu:0"], engine=eng), g, d


def test_exponential_decay_matches_reference_formula():
    for it, jobs, n, lr, mj in [(0, 1, 10, 8e-5, True), (3, 2, 10, 1e-3, True), (9, 4, 10, 1e-3, True), (5, 2, 10, 0.05, False)]:
        assert math.isclose(exponential_decay(it, jobs, n, lr, mj), O.exponential_decay(it, jobs, n, lr, mj))


def test_unknown_g_type_raises_value_error():
    cfg = small_cfg()
    with pytest.raises(ValueError, match="Unrecognized G type"):
        GAN_RNN(None, args_for(cfg, 2, g_type="cnn"), ["cpu:0"], engine=object())


def test_scalars_are_plumbed_to_the_engine():
    cfg = small_cfg()
    m, _, _ = _model(cfg, 2, g_learning_rate=3e-4, d_learning_rate=2e-3, init_mse_weight=7.0)
    o = m.engine.o
    assert math.isclose(o.g_learning_rate, np.float32(3e-4)) and math.isclose(o.d_learning_rate, np.float32(2e-3))
    assert o.mse_lambda == 7.0
    m.d_real = 0.9; m.assign("d_fake", 0.1); m.g_learning_rate = 1e-5
    assert math.isclose(o.d_real, np.float32(0.9)) and math.isclose(o.d_fake, np.float32(0.1))
    assert math.isclose(m.g_learning_rate, 1e-5)


def test_train_one_iteration_equals_reference_schedule():
    cfg = small_cfg()
    B, T = 3, 5
    m, g, d = _model(cfg, B, seed=1, disc_updates=1, gen_updates=2)
    ref = O.GanRnnOracle(cfg, g, d, batch_size=B, g_learning_rate=float(np.float32(8e-5)), d_learning_rate=float(np.float32(1e-3)))
    batches = [rand_batch(cfg, B, T, 1), rand_batch(cfg, 2, T, 2), rand_batch(cfg, B, T, 3)]
    q = queue.Queue()
    for x, lab, ln in batches:
        q.put([None, x, lab, ln])
    got = train_one_iteration(None, m, 3, 0, q)
    want = O.train_one_iteration(ref, batches, 1, 2)
    assert np.allclose(got, want, rtol=1e-6)
    for k in ref.g:
        assert np.allclose(m.engine.o.g[k], ref.g[k], rtol=1e-12)
    ev = eval_one_iteration(None, m, 1, 0, [[None] + list(batches[0])])
    assert len(ev) == 7 and math.isclose(ev[2], ev[0] + ev[1], rel_tol=1e-6)


def test_noise_is_drawn_per_call_with_std():
    cfg = small_cfg()
    m, _, _ = _model(cfg, 4, init_disc_noise_std=0.5)
    a, b = m._draw_noise(), m._draw_noise()
    assert a.shape == (4, 1, cfg.output_dim) and not np.allclose(a.numpy(), b.numpy())
    assert 0.2 < float(a.std()) < 0.9
    m.disc_noise_std = 0.0
    assert m._draw_noise() is None


def test_checkpoint_roundtrip_and_max_to_keep(tmp_path):
    cfg = small_cfg()
    B, T = 2, 4
    m, _, _ = _model(cfg, B, seed=2)
    x, lab, ln = rand_batch(cfg, B, T, 7)
    for step in range(12):
        if step < 2:
            m.d_step(x, lab, ln); m.g_step(x, lab, ln)
        m.save(str(tmp_path), step)
    import glob
    assert len(glob.glob(str(tmp_path / "GAN_RNN-*.npz"))) == 10            # Saver(max_to_keep=10)
    m2, _, _ = _model(cfg, B, seed=3)
    assert m2.load(str(tmp_path))
    for k in m.engine.o.g:
        assert np.allclose(m2.engine.o.g[k], m.engine.o.g[k].astype(np.float32), rtol=1e-6)
        assert np.allclose(m2.engine.o.adam_v[k], m.engine.o.adam_v[k].astype(np.float32), rtol=1e-6)
    assert m2.engine.o.adam_t == 2
    m3, _, _ = _model(cfg, B, seed=4)
    assert m3.load(str(tmp_path), moving_average=True)                       # EMA shadow as variables
    for k in m.engine.o.g:
        assert np.allclose(m3.engine.o.g[k], m.engine.o.g_ema[k].astype(np.float32), rtol=1e-6)
    assert not m3.load(str(tmp_path / "nope"))


def test_get_vars_asserts_prefixes():
    cfg = small_cfg("res_lstm_l")
    m, g, d = _model(cfg, 2)
    gv, dv = m.get_vars()
    assert set(gv) == set(g) and set(dv) == set(d)
    assert all(k.startswith("g_") for k in gv) and all(k.startswith("d_") for k in dv)
