"""Parity of the HIP path (through the C ABI) against the fp64 CPU oracle on seeded inputs.

Tolerances: the north star asks for <= 1e-3 relative on the G/D losses and on the enhanced-MFCC
L1; fp32 kernels against the fp64 oracle are expected to land around 1e-6..1e-5, so the tests
assert 1e-4 on losses/outputs and 2e-3 relative-L2 on every gradient / updated tensor."""
import numpy as np
import pytest
import torch

from oracle import rsrgan_oracle as O
from tests.helpers import (NET_D, NET_G, build_hip_pair, rand_batch, rel_err, small_cfg, split_flat)

pytestmark = pytest.mark.gpu

LOSS_RTOL = 1e-4
GRAD_RTOL = 2e-3
SCHEDULES = [0, 1, 3]     # 0 = layer-sequential, 1 = RSRGAN_FLAG_WAVEFRONT, 3 = wavefront replayed as hipGraphs


def _check_grads(model, net, want, tag):
    got = split_flat(model.engine.get_grads(net).cpu().numpy(), model.engine.tensor_table(net))
    bad = []
    for k in want:
        scale = max(np.abs(want[k]).max(), 1e-12)
        e = rel_err(got[k], want[k])
        if not (e < GRAD_RTOL or np.abs(got[k] - want[k]).max() < 1e-6 * max(scale, 1.0)):
            bad.append((k, e))
    assert not bad, "%s gradient mismatch: %s" % (tag, bad)


@pytest.mark.parametrize("flags", SCHEDULES)
@pytest.mark.parametrize("g_type", ["lstm", "res_lstm_l", "res_lstm_base"])
@pytest.mark.parametrize("ragged", [False, True])
def test_forward_and_gradients_small(g_type, ragged, flags):
    cfg = small_cfg(g_type)
    B, T = 5, 7
    model, oracle = build_hip_pair(cfg, B, T, seed=3, flags=flags)
    x, lab, ln = rand_batch(cfg, B, T, seed=11, ragged=ragged)
    y = model.forward(x, ln)
    y_ref = oracle.forward(x, ln)
    assert np.abs(y - y_ref).mean() / np.abs(y_ref).mean() < LOSS_RTOL          # enhanced-MFCC L1
    assert np.abs(y - y_ref).max() < 1e-4
    rng = np.random.default_rng(5)
    nr = rng.normal(0, 0.05, (B, 1, cfg.output_dim)).astype(np.float32)
    nf = rng.normal(0, 0.05, (B, 1, cfg.output_dim)).astype(np.float32)
    # D tower: losses + d_vars gradients
    got = model.engine.d_backward(x, lab, ln, nr, nf, train=True, apply=False).cpu().numpy()
    want, wg = oracle.d_tower(x.astype(np.float64), lab.astype(np.float64), ln, nr.astype(np.float64), nf.astype(np.float64))
    assert np.allclose(got, want, rtol=LOSS_RTOL), (got, want)
    _check_grads(model, NET_D, wg, "D")
    # G tower
    got = model.engine.g_backward(x, lab, ln, nf, train=True, reuse=False, apply=False).cpu().numpy()
    want, wg, _ = oracle.g_tower(x.astype(np.float64), lab.astype(np.float64), ln, nf.astype(np.float64))
    assert np.allclose(got, want, rtol=LOSS_RTOL, atol=1e-7), (got, want)
    _check_grads(model, NET_G, wg, "G")


@pytest.mark.parametrize("flags", SCHEDULES)
def test_steps_update_weights_like_oracle(flags):
    """1 D-update + 2 G-updates (the shipped 1:2 schedule, run_gan_rnn_placeholder.sh:129-130),
    first G-run reusing the D-run's generator forward; then compare every variable, Adam slot
    and EMA shadow."""
    cfg = small_cfg("lstm")
    B, T = 6, 9
    model, oracle = build_hip_pair(cfg, B, T, seed=4, flags=flags, l2_scale=1e-3, g_learning_rate=1e-3, d_learning_rate=5e-2)
    x, lab, ln = rand_batch(cfg, B, T, seed=12, ragged=True)
    a = model.d_step(x, lab, ln); b = oracle.d_step(x, lab, ln)
    assert np.allclose(np.ravel(a), np.ravel(b), rtol=LOSS_RTOL)
    for i in range(2):
        a = model.g_step(x, lab, ln, reuse_g_forward=(i == 0)); b = oracle.g_step(x, lab, ln)
        assert np.allclose(np.ravel(a), np.ravel(b), rtol=LOSS_RTOL), (i, a, b)
    gv, dv = model.get_vars()
    for k in oracle.g:
        assert rel_err(gv[k], oracle.g[k]) < 1e-4, k
    for k in oracle.d:
        assert rel_err(dv[k], oracle.d[k]) < 1e-4, k
    eng = model.engine
    m = split_flat(eng.get_params(NET_G, "adam_m").cpu().numpy(), eng.tensor_table(NET_G))
    v = split_flat(eng.get_params(NET_G, "adam_v").cpu().numpy(), eng.tensor_table(NET_G))
    e = split_flat(eng.get_params(NET_G, "ema").cpu().numpy(), eng.tensor_table(NET_G))
    for k in oracle.g:
        assert rel_err(m[k], oracle.adam_m[k]) < GRAD_RTOL, k
        assert rel_err(v[k], oracle.adam_v[k]) < 2 * GRAD_RTOL, k
        assert rel_err(e[k], oracle.g_ema[k]) < 1e-4, k
    assert eng.get_scalar("adam_step") == 2


def test_clip_by_norm_engages():
    """Large gradients: per-tensor clip at 15 must bound the SGD step (gan_rnn_placeholder.py:178-182)."""
    cfg = small_cfg("lstm")
    B, T = 4, 6
    model, oracle = build_hip_pair(cfg, B, T, seed=8, d_learning_rate=1e-2)
    x, lab, ln = rand_batch(cfg, B, T, seed=13)
    lab = lab * 300.0                                   # huge logits error -> gradient norms >> 15
    model.d_step(x, lab, ln); oracle.d_step(x, lab, ln)
    _, dv = model.get_vars()
    for k in oracle.d:
        assert rel_err(dv[k], oracle.d[k]) < 1e-4, k


def test_eval_fetch_does_not_update():
    cfg = small_cfg("lstm")
    model, oracle = build_hip_pair(cfg, 4, 6, seed=9)
    x, lab, ln = rand_batch(cfg, 4, 6, seed=14)
    g0, d0 = model.get_vars()
    a = model.d_step(x, lab, ln, train=False); b = oracle.d_step(x, lab, ln, train=False)
    assert np.allclose(np.ravel(a), np.ravel(b), rtol=LOSS_RTOL)
    a = model.g_step(x, lab, ln, train=False); b = oracle.g_step(x, lab, ln, train=False)
    assert np.allclose(np.ravel(a), np.ravel(b), rtol=LOSS_RTOL)
    g1, d1 = model.get_vars()
    assert all(np.array_equal(g0[k], g1[k]) for k in g0) and all(np.array_equal(d0[k], d1[k]) for k in d0)


@pytest.mark.parametrize("flags", SCHEDULES)
def test_edge_lengths_and_batch_sizes(flags):
    """B not a multiple of 16/32, T=1, a row of length 1 and full-length rows."""
    cfg = small_cfg("lstm")
    for B, T in ((1, 1), (17, 3), (33, 2), (70, 2)):
        model, oracle = build_hip_pair(cfg, B, T, seed=B, flags=flags)
        x, lab, ln = rand_batch(cfg, B, T, seed=15 + B)
        ln[-1] = 1
        got = model.engine.g_backward(x, lab, ln, None, train=True, reuse=False, apply=False).cpu().numpy()
        want, wg, _ = oracle.g_tower(x.astype(np.float64), lab.astype(np.float64), ln)
        assert np.allclose(got, want, rtol=LOSS_RTOL, atol=1e-7), (B, T, got, want)
        _check_grads(model, NET_G, wg, "G B=%d T=%d" % (B, T))


@pytest.mark.parametrize("flags", SCHEDULES)
@pytest.mark.parametrize("g_type", ["lstm", "res_lstm_l"])
def test_reference_true_shapes(g_type, flags):
    """The reference's hard-coded sizes (G 3x760/p280 or 4x760/p257, D 2x256/p40) at B=4, T=6."""
    cfg = O.NetCfg() if g_type == "lstm" else O.NetCfg.res_lstm_l()
    B, T = 4, 6
    model, oracle = build_hip_pair(cfg, B, T, seed=21, flags=flags)
    x, lab, ln = rand_batch(cfg, B, T, seed=22, ragged=True)
    got = model.engine.d_backward(x, lab, ln, None, None, train=True, apply=False).cpu().numpy()
    want, wg = oracle.d_tower(x.astype(np.float64), lab.astype(np.float64), ln)
    assert np.allclose(got, want, rtol=LOSS_RTOL), (got, want)
    _check_grads(model, NET_D, wg, "D")
    got = model.engine.g_backward(x, lab, ln, None, train=True, reuse=True, apply=False).cpu().numpy()
    want, wg, y_ref = oracle.g_tower(x.astype(np.float64), lab.astype(np.float64), ln)
    assert np.allclose(got, want, rtol=LOSS_RTOL), (got, want)
    _check_grads(model, NET_G, wg, "G")
    y = model.forward(x, ln)
    assert np.abs(y - y_ref).mean() / np.abs(y_ref).mean() < LOSS_RTOL


@pytest.mark.parametrize("flags", SCHEDULES)
def test_train_one_iteration_matches_oracle(flags):
    from rsrgan_amd import train_one_iteration
    cfg = small_cfg("lstm")
    B, T = 4, 5
    model, oracle = build_hip_pair(cfg, B, T, seed=31, flags=flags, disc_updates=1, gen_updates=2)
    batches = [rand_batch(cfg, B, T, seed=40), rand_batch(cfg, 3, T, seed=41), rand_batch(cfg, B, T, seed=42)]
    queue = [[None, x, lab, ln] for x, lab, ln in batches]
    got = train_one_iteration(None, model, len(queue), 0, queue)
    want = O.train_one_iteration(oracle, batches, disc_updates=1, gen_updates=2)
    assert np.allclose(got, want, rtol=LOSS_RTOL), (got, want)


def test_checkpoint_roundtrip(tmp_path):
    cfg = small_cfg("lstm")
    model, _ = build_hip_pair(cfg, 4, 5, seed=51)
    x, lab, ln = rand_batch(cfg, 4, 5, seed=52)
    model.d_step(x, lab, ln); model.g_step(x, lab, ln)
    model.save(str(tmp_path), 7)
    g0, d0 = model.get_vars()
    ref = model.g_step(x, lab, ln, train=False)
    model2, _ = build_hip_pair(cfg, 4, 5, seed=99)
    assert model2.load(str(tmp_path))
    g1, d1 = model2.get_vars()
    assert all(np.array_equal(g0[k], g1[k]) for k in g0) and all(np.array_equal(d0[k], d1[k]) for k in d0)
    assert model2.engine.get_scalar("adam_step") == 1
    assert np.allclose(np.ravel(model2.g_step(x, lab, ln, train=False)), np.ravel(ref), rtol=1e-6)
    assert not model2.load(str(tmp_path / "missing"))


@pytest.mark.parametrize("flags", SCHEDULES)
@pytest.mark.parametrize("g_type", ["lstm", "res_lstm_base"])
def test_num_proj_none_layers(g_type, flags):
    """tf.contrib.rnn.LSTMCell(num_proj=None): m = h (BASELINE.json's '2x512' wording); no projection launch."""
    cfg = small_cfg(g_type, g_proj=0, d_proj=0)
    B, T = 6, 7
    model, oracle = build_hip_pair(cfg, B, T, seed=71, flags=flags)
    assert not any("projection" in n for n, _, _ in model.engine.tensor_table(NET_G))
    x, lab, ln = rand_batch(cfg, B, T, seed=72, ragged=True)
    y = model.forward(x, ln)
    assert np.abs(y - oracle.forward(x, ln)).max() < 1e-4
    got = model.engine.d_backward(x, lab, ln, None, None, train=True, apply=False).cpu().numpy()
    want, wg = oracle.d_tower(x.astype(np.float64), lab.astype(np.float64), ln)
    assert np.allclose(got, want, rtol=LOSS_RTOL), (got, want)
    _check_grads(model, NET_D, wg, "D")
    got = model.engine.g_backward(x, lab, ln, None, train=True, reuse=True, apply=False).cpu().numpy()
    want, wg, _ = oracle.g_tower(x.astype(np.float64), lab.astype(np.float64), ln)
    assert np.allclose(got, want, rtol=LOSS_RTOL, atol=1e-7), (got, want)
    _check_grads(model, NET_G, wg, "G")
    a = model.d_step(x, lab, ln); b = oracle.d_step(x, lab, ln)
    assert np.allclose(np.ravel(a), np.ravel(b), rtol=LOSS_RTOL)
    a = model.g_step(x, lab, ln); b = oracle.g_step(x, lab, ln)
    assert np.allclose(np.ravel(a), np.ravel(b), rtol=LOSS_RTOL)


def test_baseline_named_network():
    """BASELINE.json configs[1]: 2-layer 512-unit LSTM generator (no projection) + DNN discriminator, B=4, T=6."""
    from rsrgan_amd import GAN_RNN
    from tests.helpers import args_for, rand_params
    cfg = O.NetCfg(g_type="lstm", g_layers=2, g_cells=512, g_proj=0, d_type="dnn", d_layers=4, d_cells=1024)
    B, T = 4, 6
    g, d = rand_params(cfg, 81)
    m = GAN_RNN(None, args_for(cfg, B), ["gpu:0"], max_frames=T,
                net_overrides=dict(g_layers=2, g_cells=512, g_proj=0, d_type="dnn", d_layers=4, d_cells=1024, flags=1))
    m.set_vars(g, d)
    o = O.GanRnnOracle(cfg, g, d, batch_size=B, g_learning_rate=float(np.float32(8e-5)), d_learning_rate=float(np.float32(1e-3)))
    x, lab, ln = rand_batch(cfg, B, T, 82, ragged=True)
    got = m.engine.d_backward(x, lab, ln, train=True, apply=False).cpu().numpy()
    want, wg = o.d_tower(x.astype(np.float64), lab.astype(np.float64), ln)
    assert np.allclose(got, want, rtol=LOSS_RTOL), (got, want)
    _check_grads(m, NET_D, wg, "D")
    got = m.engine.g_backward(x, lab, ln, train=True, reuse=True, apply=False).cpu().numpy()
    want, wg, _ = o.g_tower(x.astype(np.float64), lab.astype(np.float64), ln)
    assert np.allclose(got, want, rtol=LOSS_RTOL), (got, want)
    _check_grads(m, NET_G, wg, "G")


def test_outer_loop_on_gpu(tmp_path):
    """train() (decay, accept/reject checkpoint) then decode() -> ark on the HIP engine with scp/ark data."""
    from rsrgan_amd import run_gan_rnn as R
    from rsrgan_amd.io import ArkReader, ArkWriter
    rng = np.random.default_rng(0)
    din, dout = 5, 3

    def data(n, tag):
        wi, wl = ArkWriter(str(tmp_path / (tag + "_in.scp"))), ArkWriter(str(tmp_path / (tag + "_lab.scp")))
        for i in range(n):
            T = int(rng.integers(6, 10))
            wi.write_next_utt(str(tmp_path / (tag + "_in.ark")), "%s%02d" % (tag, i), rng.standard_normal((T, din)))
            wl.write_next_utt(str(tmp_path / (tag + "_lab.ark")), "%s%02d" % (tag, i), rng.standard_normal((T, dout)))
        wi.close(); wl.close()
        return str(tmp_path / (tag + "_in.scp")), str(tmp_path / (tag + "_lab.scp"))
    tr, cv, te = data(8, "tr"), data(4, "cv"), data(2, "te")
    np.savez(tmp_path / "train_cmvn.npz", mean_inputs=np.zeros(din), stddev_inputs=np.ones(din), mean_labels=np.zeros(dout), stddev_labels=np.ones(dout))
    FLAGS, _ = R.build_parser().parse_known_args([
        "--data_dir", str(tmp_path), "--tr_inputs_scp", tr[0], "--tr_labels_scp", tr[1], "--cv_inputs_scp", cv[0], "--cv_labels_scp", cv[1],
        "--test_inputs_scp", te[0], "--input_dim", str(din), "--output_dim", str(dout), "--left_context", "1", "--right_context", "1",
        "--batch_size", "2", "--min_epoches", "1", "--max_epoches", "2", "--save_dir", str(tmp_path / "exp"), "--max_frames", "16",
        "--init_disc_noise_std", "0.05", "--g_learning_rate", "0.003"])
    ov = dict(g_layers=1, g_cells=8, g_proj=4, d_layers=1, d_cells=8, d_proj=3, flags=1)
    logs = []
    hist = R.train(FLAGS, log=logs.append, net_overrides=ov)
    assert len(hist) >= 1 and all(np.isfinite(hist)) and "Nnet Accepted" in "\n".join(logs)
    FLAGS.decode = True
    scp = R.decode(FLAGS, log=logs.append, net_overrides=ov)
    r = ArkReader(); r(scp)
    assert len(r.utt_ids) == 2
    for i in range(2):
        m = r.read_utt_data_from_index(i)
        assert m.shape[1] == dout and np.all(np.isfinite(m))


def test_profile_api_counts_gates_launches():
    """rsrgan_profile_begin/read (bench.py's live dominant-kernel timing): launch count, positive time, algorithmic FLOPs
    = sum over (layer, t) jobs of 2*N*(I+P)*4H, and no effect on the results."""
    cfg = small_cfg()
    B, T = 4, 6
    model, oracle = build_hip_pair(cfg, B, T, seed=5, flags=1)
    x, lab, ln = rand_batch(cfg, B, T, seed=6)
    ref = model.engine.d_backward(x, lab, ln, train=True, apply=False).cpu().numpy()
    model.engine.profile_begin()
    got = model.engine.d_backward(x, lab, ln, train=True, apply=False).cpu().numpy()
    n, us, flops = model.engine.profile_read()
    assert np.array_equal(ref, got)
    assert n > 0 and us > 0
    pad4 = lambda v: (v + 3) // 4 * 4
    lst = lambda n_, i, h, p: 2.0 * n_ * (pad4(i) + pad4(p)) * 4 * h
    # generator: layer 0's x-part is batched into a GEMM (not in this kernel); discriminator on 2B rows (real + fake)
    want = T * (lst(B, 0, cfg.g_cells, cfg.g_proj) + (cfg.g_layers - 1) * lst(B, cfg.g_proj, cfg.g_cells, cfg.g_proj))
    want += T * (lst(2 * B, cfg.output_dim, cfg.d_cells, cfg.d_proj) + (cfg.d_layers - 1) * lst(2 * B, cfg.d_proj, cfg.d_cells, cfg.d_proj))
    assert abs(flops - want) <= 1e-6 * want, (flops, want)
    n2, us2, _ = model.engine.profile_read()          # profiling is off again: nothing new recorded by a further step
    model.engine.d_backward(x, lab, ln, train=True, apply=False)
    assert model.engine.profile_read()[0] == n2


@pytest.mark.parametrize("g_type", ["lstm", "res_lstm_l"])
def test_long_sequences(g_type):
    """T = 300 padded frames (the reference buckets utterances up to ~1000 frames): BPTT over hundreds of steps stays within the
    tolerances of the short cases; lengths ragged, one row a single frame long."""
    cfg = small_cfg(g_type)
    B, T = 3, 300
    model, oracle = build_hip_pair(cfg, B, T, seed=41, flags=1)
    x, lab, ln = rand_batch(cfg, B, T, seed=42, ragged=True)
    ln[1] = 1
    got = model.engine.d_backward(x, lab, ln, None, None, train=True, apply=False).cpu().numpy()
    want, wg = oracle.d_tower(x.astype(np.float64), lab.astype(np.float64), ln)
    assert np.allclose(got, want, rtol=LOSS_RTOL), (got, want)
    _check_grads(model, NET_D, wg, "D long")
    got = model.engine.g_backward(x, lab, ln, None, train=True, reuse=True, apply=False).cpu().numpy()
    want, wg, y_ref = oracle.g_tower(x.astype(np.float64), lab.astype(np.float64), ln)
    assert np.allclose(got, want, rtol=LOSS_RTOL), (got, want)
    _check_grads(model, NET_G, wg, "G long")
    y = model.forward(x, ln)
    assert np.abs(y - y_ref).max() < 1e-4


@pytest.mark.parametrize("g_type", ["lstm", "res_lstm_l"])
def test_graph_replay_is_bit_identical_to_eager(g_type):
    """RSRGAN_FLAG_GRAPH: a segment runs eagerly on its first use, is captured on the second and replayed from the third; the
    kernels and their order are the same, so six iterations of the shipped 1 D + 2 G schedule (alternating T, with and without
    noise, changing scalars in between) must give bit-identical losses and variables with and without graphs."""
    cfg = small_cfg(g_type)
    B = 6
    eager, _ = build_hip_pair(cfg, B, 9, seed=21, flags=1, l2_scale=1e-3)
    graph, _ = build_hip_pair(cfg, B, 9, seed=21, flags=3, l2_scale=1e-3)
    rng = np.random.default_rng(2)
    for it in range(8):
        T = 9 if it % 3 else 7                      # two graph sets, each one used at least three times
        x, lab, ln = rand_batch(cfg, B, T, seed=50 + it, ragged=it % 2 == 1)
        nr = nf = None
        if it % 4 >= 2:
            nr = rng.normal(0, 0.05, (B, 1, cfg.output_dim)).astype(np.float32)
            nf = rng.normal(0, 0.05, (B, 1, cfg.output_dim)).astype(np.float32)
        for m in (eager, graph):
            m.engine.set_scalar("g_learning_rate", 1e-3 * (1 + it))
        outs = []
        for m in (eager, graph):
            d = np.ravel(m.engine.d_backward(x, lab, ln, nr, nf, train=True, apply=True).cpu().numpy())
            g1 = np.ravel(m.engine.g_backward(x, lab, ln, nf, train=True, reuse=True, apply=True).cpu().numpy())
            g2 = np.ravel(m.engine.g_backward(x, lab, ln, nf, train=True, reuse=False, apply=True).cpu().numpy())
            ev = np.ravel(m.engine.g_backward(x, lab, ln, nf, train=False, reuse=False).cpu().numpy())
            outs.append(np.concatenate([d, g1, g2, ev]))
        assert np.array_equal(outs[0], outs[1]), (it, outs[0], outs[1])
    for net in (NET_G, NET_D):
        a = eager.engine.get_params(net).cpu().numpy(); b = graph.engine.get_params(net).cpu().numpy()
        assert np.array_equal(a, b)
    assert np.array_equal(eager.forward(x, ln), graph.forward(x, ln))


@pytest.mark.parametrize("g_type,flags,B,T", [("lstm", 0, 6, 9), ("lstm", 1, 6, 9), ("lstm", 3, 6, 9), ("res_lstm_l", 3, 5, 7),
                                              ("lstm", 3, 33, 4)])
def test_dropout_wrapper_on_the_generator_layers(g_type, flags, B, T):
    """--keep_prob < 1 (train_gan_rnn_placeholder.py:723-728): tf.contrib.rnn.DropoutWrapper(cell, output_keep_prob) around every
    generator layer (models/lstm.py:99-102, res_lstm_l.py:96-99) -- masks per (training run, layer, t), drawn on the device inside
    k_fwd_proj / k_bwd_a2 and replayed correctly by hipGraphs (the run index lives in device memory); the oracle is fed the same
    masks (tests/helpers.py seq_dropout_mask).  Losses of D- and G-runs, every variable after 2 rounds, evaluation fetches
    undropped, a G-run never reuses the D-run's (differently masked) forward."""
    cfg = small_cfg(g_type)
    model, oracle = build_hip_pair(cfg, B, T, seed=4, flags=flags, l2_scale=1e-3, g_learning_rate=1e-3, d_learning_rate=5e-2, keep_prob=0.8)
    plain = build_hip_pair(cfg, B, T, seed=4, flags=flags, l2_scale=1e-3, g_learning_rate=1e-3, d_learning_rate=5e-2)[1]
    x, lab, ln = rand_batch(cfg, B, T, seed=12, ragged=True)
    for rnd in range(2):
        a = model.d_step(x, lab, ln); b = oracle.d_step(x, lab, ln)
        assert np.allclose(np.ravel(a), np.ravel(b), rtol=LOSS_RTOL), (rnd, a, b)
        if rnd == 0:
            assert not np.allclose(np.ravel(b)[1], np.ravel(plain.d_step(x, lab, ln))[1], rtol=1e-3)      # the masks do act
        for i in range(2):
            a = model.g_step(x, lab, ln, reuse_g_forward=(i == 0)); b = oracle.g_step(x, lab, ln)
            assert np.allclose(np.ravel(a), np.ravel(b), rtol=LOSS_RTOL), (rnd, i, a, b)
    gv, dv = model.get_vars()
    for k in oracle.g:
        assert rel_err(gv[k], oracle.g[k]) < 2e-4, k
    for k in oracle.d:
        assert rel_err(dv[k], oracle.d[k]) < 2e-4, k
    a = model.g_step(x, lab, ln, train=False); b = oracle.g_step(x, lab, ln, train=False)
    assert np.allclose(np.ravel(a)[:2], np.ravel(b)[:2], rtol=LOSS_RTOL)       # (g_adv, g_mse: the twin's fetch carries no L2 term)
    a = model.d_step(x, lab, ln, train=False); b = oracle.d_step(x, lab, ln, train=False)
    assert np.allclose(np.ravel(a), np.ravel(b), rtol=LOSS_RTOL)
    assert model.engine.device_status() == 0
