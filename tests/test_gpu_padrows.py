"""Batches that are no multiple of the persistent generator kernels' 32-row group (the shipped recipe trains with batch_size=8,
run_gan_rnn_placeholder.sh:126; decode feeds one utterance, train_gan_rnn_placeholder.py:282-285) are PADDED with rows of length 0
(csrc/model.h Bt): dynamic_rnn's masking makes such rows inert and the loss kernels leave them out of every mean, so the results are
the reference's for the caller's rows -- checked against the fp64 oracle at the caller's batch size -- while the recurrences run as
the persistent launches (launch counters prove it).  Also: what happens when the device cannot hold a persistent launch."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import rsrgan_oracle as O
from tests.helpers import NET_D, NET_G, build_hip_pair, rand_batch, rel_err
from tests.test_gpu_fullsize import RTOL, _grads, _step_against_oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("B,T,flags", [(8, 100, 3), (8, 16, 1), (1, 50, 1), (40, 9, 3), (33, 5, 1)])
def test_padded_batch_against_oracle(B, T, flags):
    _step_against_oracle(O.NetCfg(), B, T, flags, seed=500 + B)


@pytest.mark.parametrize("mode", ["1", "2"])
@pytest.mark.parametrize("net", ["lstm", "res_lstm_l"])
def test_padded_batch_with_one_persistent_direction_against_oracle(net, mode, monkeypatch):
    """Round 6 (advisor): a padded batch (batch_size 8 -> one 32-row group) with only ONE of the generator's recurrences persistent
    (RSRGAN_GPERSIST=1: forward only, =2: BPTT only; read at rsrgan_create).  The one-lane form of the persistent launches (GPersistArgs::
    nrt = 1) relies on the padding rows of the stash never being written; a launch-path recurrence writes them, so the lane is dropped
    only when both directions run persistent -- every gradient against the oracle either way."""
    monkeypatch.setenv("RSRGAN_GPERSIST", mode)
    cfg = O.NetCfg.res_lstm_l() if net == "res_lstm_l" else O.NetCfg()
    _step_against_oracle(cfg, 8, 12, 3, seed=640)


@pytest.mark.parametrize("B,T,flags", [(8, 100, 3), (8, 12, 1), (1, 30, 1)])
def test_shipped_recipe_batch_against_oracle(B, T, flags):
    """run_gan_rnn_placeholder.sh:124,126: --g_type res_lstm_l --batch_size 8 -- the residual stack, padded to one 32-row group, on the
    persistent launches (csrc/gpersist.hip RES)."""
    cfg = O.NetCfg.res_lstm_l()
    _step_against_oracle(cfg, B, T, flags, seed=600 + B)
    if flags == 1:
        model, _ = build_hip_pair(cfg, B, T, seed=600 + B, flags=1)
        x, lab, ln = rand_batch(cfg, B, T, seed=700 + B, ragged=True)
        model.engine.profile_begin()
        model.engine.d_backward(x, lab, ln, None, None, train=True, apply=False)
        model.engine.g_backward(x, lab, ln, None, train=True, reuse=True, apply=False)
        # (RSRGAN_DPIPE=1, the Python layer's default: the D-run's forward is k_glstm_fwd_dt = kind 3, else k_glstm_fwd = kind 1)
        n = (model.engine.profile_read_kind(1)[0] + model.engine.profile_read_kind(3)[0], model.engine.profile_read_kind(2)[0])
        model.engine.profile_read()
        assert n == (1, 1), "res_lstm_l did not take the persistent launches: %r" % (n,)
        assert model.engine.device_status() == 0


def test_padded_batch_runs_the_persistent_launches_and_takes_noise():
    cfg = O.NetCfg()
    B, T = 8, 12
    model, oracle = build_hip_pair(cfg, B, T, seed=21, flags=1)
    x, lab, ln = rand_batch(cfg, B, T, seed=22, ragged=True)
    rng = np.random.default_rng(23)
    nr = (0.3 * rng.standard_normal((B, 1, cfg.output_dim))).astype(np.float32)
    nf = (0.3 * rng.standard_normal((B, 1, cfg.output_dim))).astype(np.float32)
    model.engine.profile_begin()
    got = model.engine.d_backward(x, lab, ln, nr, nf, train=True, apply=False).cpu().numpy()
    want, wg = oracle.d_tower(x.astype(np.float64), lab.astype(np.float64), ln, nr.astype(np.float64), nf.astype(np.float64))
    assert np.allclose(got, want, rtol=RTOL), (got, want)
    gd = _grads(model, NET_D)
    for k in wg:
        assert rel_err(gd[k], wg[k]) < 2e-3, ("D", k, rel_err(gd[k], wg[k]))
    got = model.engine.g_backward(x, lab, ln, nf, train=True, reuse=True, apply=False).cpu().numpy()
    want, wg, _ = oracle.g_tower(x.astype(np.float64), lab.astype(np.float64), ln, nf.astype(np.float64))
    assert np.allclose(got, want, rtol=RTOL), (got, want)
    gg = _grads(model, NET_G)
    for k in wg:
        assert rel_err(gg[k], wg[k]) < 2e-3, ("G", k, rel_err(gg[k], wg[k]))
    n_fwd, n_bwd = model.engine.profile_read_kind(1)[0] + model.engine.profile_read_kind(3)[0], model.engine.profile_read_kind(2)[0]
    model.engine.profile_read()
    assert (n_fwd, n_bwd) == (1, 1), "the persistent generator launches did not run on the padded batch (%d, %d)" % (n_fwd, n_bwd)
    assert model.engine.device_status() == 0
    # decode: one utterance, output rows of the caller's batch only
    one, o1 = build_hip_pair(cfg, 1, 40, seed=21, flags=1)
    x1, _, l1 = rand_batch(cfg, 1, 40, seed=24)
    y = one.forward(x1, l1)
    assert y.shape == (1, 40, cfg.output_dim)
    y_ref = o1.forward(x1.astype(np.float64), l1)
    assert np.abs(y - y_ref).mean() / np.abs(y_ref).mean() < RTOL


WORKER = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %r)
from oracle import rsrgan_oracle as O
from tests.helpers import build_hip_pair, rand_batch
cfg = O.NetCfg()
B, T = 64, 12
model, oracle = build_hip_pair(cfg, B, T, seed=31, flags=int(os.environ.get("RSRGAN_TEST_FLAGS", "3")))
x, lab, ln = rand_batch(cfg, B, T, seed=32, ragged=True)
out = {"d": [], "g": []}
model.engine.profile_begin() if os.environ.get("RSRGAN_TEST_FLAGS") == "1" else None
for it in range(3):      # (the first pass without the updates: a failed launch must not reach the variables)
    out["d"].append([float(v) for v in np.ravel(model.d_step(x, lab, ln, train=it > 0))])
    out["g"].append([float(v) for v in np.ravel(model.g_step(x, lab, ln, train=it > 0, reuse_g_forward=it > 0))])
    out.setdefault("status", []).append(int(model.engine.device_status()))
if os.environ.get("RSRGAN_TEST_FLAGS") == "1":
    out["n_gp"] = int(model.engine.profile_read_kind(1)[0] + model.engine.profile_read_kind(2)[0] + model.engine.profile_read_kind(3)[0]); model.engine.profile_read()
# after the passes: the variables put back (a failed launch's update has poisoned them), one evaluation on whatever path the handle
# is on now against the oracle at the same variables
from tests.helpers import rand_params
g0, d0 = rand_params(cfg, 31)
model.set_vars(g0, d0)
ev_d = np.ravel(model.d_step(x, lab, ln, train=False)); ev_g = np.ravel(model.g_step(x, lab, ln, train=False))
out["status"].append(int(model.engine.device_status()))
out["recovered"] = bool(np.allclose(ev_d, np.ravel(oracle.d_step(x, lab, ln, train=False)), rtol=1e-3) and
                        np.allclose(ev_g, np.ravel(oracle.g_step(x, lab, ln, train=False)), rtol=1e-3))
want_d = np.ravel(oracle.d_step(x, lab, ln)); want_g = np.ravel(oracle.g_step(x, lab, ln))
out["ok"] = bool(np.allclose(out["d"][1], want_d, rtol=1e-3) and np.allclose(out["g"][1], want_g, rtol=1e-3))
import torch
out["cus"] = int(torch.cuda.get_device_properties(0).multi_processor_count)
print("RESULT " + json.dumps(out))
""" % ROOT


def _worker(env):
    e = dict(os.environ); e.update(env)
    p = subprocess.run([sys.executable, "-c", WORKER], capture_output=True, text=True, env=e, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


@pytest.mark.parametrize("mask", [{"HSA_CU_MASK": "0:0-127"}, {"ROC_GLOBAL_CU_MASK": "0x" + "f" * 32}])
def test_half_the_device_falls_back_without_time_outs(mask):
    """With half of the CUs masked away the 228-workgroup generator launches cannot be resident at once.  rsrgan_create asks the device
    (csrc/gpersist.hip resident_probe: multiProcessorCount does not know about masks) and leaves the hand-off rings unallocated: the
    steps run on the launch-per-phase path, agree with the oracle, and no bounded wait ever expires (device_status 0, no 1 s stalls)."""
    import time
    t0 = time.time()
    r = _worker(dict(mask, RSRGAN_TEST_FLAGS="1"))
    dt = time.time() - t0
    assert r["ok"] and r["recovered"] and r["status"] == [0, 0, 0, 0], r
    full = _worker({"RSRGAN_TEST_FLAGS": "1"})
    assert full["ok"] and full["recovered"] and full["status"] == [0, 0, 0, 0] and full["n_gp"] == 6, full      # (evaluation pass: the D-run's forward and the G-run's recomputed one; then forward + BPTT per training iteration)
    if r["n_gp"] != 0:
        pytest.skip("the CU mask %r is not honoured in this environment (the persistent launches ran: %d)" % (mask, r["n_gp"]))
    assert dt < 120, dt


def test_probe_verdict_no_selects_the_launch_path():
    """The fallback itself, independent of whether this environment honours CU masks: RSRGAN_RESIDENT_CAP=128 makes resident_probe
    answer what it answers on half a device (128 workgroup slots < the 228 of the generator launches, the discriminator's 64 fit):
    rsrgan_create leaves the generator's hand-off rings unallocated, the steps run on the launch-per-phase path (no persistent
    generator launch is bracketed), agree with the oracle, and no bounded wait expires."""
    r = _worker({"RSRGAN_RESIDENT_CAP": "128", "RSRGAN_TEST_FLAGS": "1"})
    assert r["n_gp"] == 0, r
    assert r["ok"] and r["recovered"] and r["status"] == [0, 0, 0, 0], r
    r = _worker({"RSRGAN_RESIDENT_CAP": "16", "RSRGAN_TEST_FLAGS": "3"})        # neither net's persistent launches fit; graph replay
    assert r["ok"] and r["recovered"] and r["status"] == [0, 0, 0, 0], r


def test_a_failed_persistent_launch_disables_the_path_for_the_handle():
    """after a reported failure the handle must stop trying (every further step would spin into the same time-out).  With
    the probe switched off (RSRGAN_RESIDENT_PROBE=0 trusts multiProcessorCount, which a CU mask does not change) a masked device
    accepts the launches at rsrgan_create and one of them reports a failed bounded wait (on this pool: k_glstm_bwd of the first
    training pass, code 0x10000 + workgroup); every pass after the report must run on the launch path, status 0, finite losses."""
    r = _worker({"HSA_CU_MASK": "0:0-127", "RSRGAN_RESIDENT_PROBE": "0", "RSRGAN_TEST_FLAGS": "3"})
    bad = [i for i, v in enumerate(r["status"]) if v != 0]
    if not bad:
        pytest.skip("the CU mask is not honoured in this environment (no launch failed)")
    assert bad[0] < 2 and r["status"][bad[0]] >= 0x10000, r["status"]
    assert all(v == 0 for v in r["status"][bad[0] + 1:]), r["status"]          # no further time-outs: the handle has left the path
    # (the failed pass was a training pass: its update ran on the poisoned gradients, as the header says of a failed launch; with the
    #  variables put back the handle computes the oracle's values on the launch path)
    assert r["recovered"], r
