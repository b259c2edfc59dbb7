"""Error surface of the C ABI (include/rsrgan.h): every status code that can be provoked safely, through ctypes.
Reference behaviour: ValueError on an unknown g_type (models/gan_rnn_placeholder.py:131-132), short / oversize feeds are the
caller's error (scripts/train_gan_rnn_placeholder.py:69-70); nothing may throw or crash across the boundary."""
import ctypes as C

import numpy as np
import pytest
import torch

from rsrgan_amd import _lib
from tests.helpers import build_hip_pair, rand_batch, small_cfg

pytestmark = pytest.mark.gpu
OK, INVALID, HIP, NO_DEVICE, STATE = 0, -1, -2, -3, -4


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def test_invalid_arguments_are_reported_not_crashed():
    lib = _lib.load()
    cfg = _lib.RsrganCfg()
    assert lib.rsrgan_default_cfg(99, C.byref(cfg)) == INVALID and b"Unrecognized G type" in lib.rsrgan_last_error()
    assert lib.rsrgan_default_cfg(0, None) == INVALID
    assert lib.rsrgan_default_cfg(0, C.byref(cfg)) == OK
    h = C.c_void_p()
    assert lib.rsrgan_create(None, C.c_uint64(1), C.byref(h)) == INVALID
    cfg.batch_size = 0
    assert lib.rsrgan_create(C.byref(cfg), C.c_uint64(1), C.byref(h)) == INVALID and b"invalid sizes" in lib.rsrgan_last_error()
    cfg.batch_size = 2; cfg.max_frames = 4; cfg.g_type = 7
    assert lib.rsrgan_create(C.byref(cfg), C.c_uint64(1), C.byref(h)) == INVALID and b"Unrecognized G type" in lib.rsrgan_last_error()
    cfg.g_type = 0; cfg.d_type = 5
    assert lib.rsrgan_create(C.byref(cfg), C.c_uint64(1), C.byref(h)) == INVALID
    cfg.d_type = 0; cfg.g_proj = 4000
    assert lib.rsrgan_create(C.byref(cfg), C.c_uint64(1), C.byref(h)) == INVALID
    for f in (lib.rsrgan_destroy,):
        assert f(None) == INVALID
    assert lib.rsrgan_apply(None, 0, None) == INVALID
    out = C.c_double()
    assert lib.rsrgan_get_scalar(None, 0, C.byref(out)) == INVALID


def test_call_sequence_and_shape_errors():
    cfg = small_cfg("lstm")
    B, T = 3, 5
    model, _ = build_hip_pair(cfg, B, T, seed=1, flags=3)
    e = model.engine
    lib, h = e.lib, e.h
    x, lab, ln = rand_batch(cfg, B, T, seed=2)
    xd = torch.from_numpy(x).cuda(); ld = torch.from_numpy(lab).cuda(); lnd = torch.from_numpy(ln).cuda()
    out = torch.zeros(4, device="cuda")
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    # apply without gradients; reuse without a forward
    assert lib.rsrgan_apply(h, 0, s) == STATE and lib.rsrgan_apply(h, 1, s) == STATE and lib.rsrgan_apply(h, 5, s) == INVALID
    assert lib.rsrgan_g_step(h, _ptr(xd), _ptr(ld), _ptr(lnd), T, None, _ptr(out), 1, 1, s) == STATE
    assert b"reuse_g_forward" in lib.rsrgan_last_error()
    # T outside (0, max_frames], null inputs / labels / lengths
    assert lib.rsrgan_d_step(h, _ptr(xd), _ptr(ld), _ptr(lnd), T + 1, None, None, _ptr(out), 1, s) == INVALID
    assert lib.rsrgan_d_step(h, _ptr(xd), _ptr(ld), _ptr(lnd), 0, None, None, _ptr(out), 1, s) == INVALID
    assert lib.rsrgan_d_step(h, None, _ptr(ld), _ptr(lnd), T, None, None, _ptr(out), 1, s) == INVALID
    assert lib.rsrgan_d_step(h, _ptr(xd), None, _ptr(lnd), T, None, None, _ptr(out), 1, s) == INVALID
    assert lib.rsrgan_d_step(h, _ptr(xd), _ptr(ld), None, T, None, None, _ptr(out), 1, s) == INVALID
    assert lib.rsrgan_forward_g(h, _ptr(xd), _ptr(lnd), T, None, s) == INVALID
    # unknown scalar / tensor index / bucket index / buffer
    assert lib.rsrgan_set_scalar(h, 99, C.c_double(1.0)) == INVALID
    assert lib.rsrgan_tensor_info(h, 0, 999, None, 0, None, None, None) == INVALID
    off, cnt = C.c_int64(), C.c_int64()
    assert lib.rsrgan_grad_bucket_info(h, 0, 99, C.byref(off), C.byref(cnt)) == INVALID
    assert lib.rsrgan_get_params(h, 0, 9, _ptr(out), s) == INVALID
    # the handle is still healthy after all of that
    got = np.ravel(model.d_step(x, lab, ln))
    assert np.all(np.isfinite(got))
    # the Python layer refuses shapes the library would read out of bounds
    with pytest.raises(ValueError):
        e.d_backward(x[:-1], lab[:-1], ln[:-1])
    with pytest.raises(ValueError):
        e.d_backward(x, lab[:, :-1], ln)
    with pytest.raises(ValueError):
        e.forward_g(x[:, :, :-1], ln)
    with pytest.raises(ValueError):
        e.forward_g(np.zeros((B, T + 1, cfg.input_dim), np.float32), ln)


def test_supervised_trainer_has_no_discriminator_step():
    from rsrgan_amd.trainer import RNNTrainer
    from tests.helpers import args_for, overrides
    cfg = small_cfg("lstm")
    m = RNNTrainer(None, args_for(cfg, 3), ["gpu:0"], max_frames=5, net_overrides=overrides(cfg))
    x, lab, ln = rand_batch(cfg, 3, 5, seed=2)
    with pytest.raises(_lib.RsrganError) as ei:
        m.engine.d_backward(x, lab, ln)
    assert ei.value.code == STATE if hasattr(ei.value, "code") else True
