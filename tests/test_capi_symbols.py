"""The C-ABI library loads and exports every symbol include/rsrgan.h declares (no compute: no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "rsrgan.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rsrgan_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from rsrgan_amd import _lib
    lib = _lib.load()
    declared = _header_functions()
    assert len(declared) >= 20
    missing = [f for f in declared if not hasattr(lib, f)]
    assert not missing, missing
    assert sorted(_lib.SYMBOLS) == declared           # the Python binding lists exactly the header


def test_cfg_struct_matches_header_defaults():
    from rsrgan_amd import _lib
    lib = _lib.load()
    cfg = _lib.RsrganCfg()
    assert lib.rsrgan_default_cfg(0, C.byref(cfg)) == 0
    # models/lstm.py:43-45 ; models/discriminator_lstm.py:26-28 ; gan_rnn_placeholder.py:70-71
    assert (cfg.g_layers, cfg.g_cells, cfg.g_proj) == (3, 760, 280)
    assert (cfg.d_layers, cfg.d_cells, cfg.d_proj) == (2, 256, 40)
    assert cfg.clip_norm == 15.0 and abs(cfg.ema_decay - 0.9999) < 1e-7 and abs(cfg.lrelu_alpha - 0.3) < 1e-7
    assert lib.rsrgan_default_cfg(1, C.byref(cfg)) == 0
    assert (cfg.g_layers, cfg.g_cells, cfg.g_proj) == (4, 760, 257)     # models/res_lstm_l.py:43-45
    assert lib.rsrgan_default_cfg(9, C.byref(cfg)) < 0                    # ValueError('Unrecognized G type')


def test_no_cpu_fallback():
    """Without a GPU the product path must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rsrgan_amd import GAN_RNN, _lib
    from tests.helpers import args_for, small_cfg
    with pytest.raises(_lib.RsrganError):
        GAN_RNN(None, args_for(small_cfg(), 2), ["gpu:0"], max_frames=4)
    cfg = _lib.RsrganCfg()
    lib = _lib.load()
    lib.rsrgan_default_cfg(0, C.byref(cfg))
    h = C.c_void_p()
    assert lib.rsrgan_create(C.byref(cfg), 1, C.byref(h)) == -3          # RSRGAN_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.rsrgan_last_error()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "rsrgan_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                assert "oracle" not in open(os.path.join(dp, f)).read().replace("no CPU fallback", ""), f
