"""rsrgan_amd/summary.py: TensorBoard event files as tf.summary.FileWriter writes them (TFRecord framing with masked crc32c, Event /
Summary / HistogramProto by hand), and the outer loops' add_summary call sites with a recording stand-in for the model."""
import glob
import os

import numpy as np

from rsrgan_amd import summary as S


def test_crc32c_known_answers_and_mask():
    # RFC 3720 B.4 test vectors + the classic check value
    assert S.crc32c(b"123456789") == 0xE3069283
    assert S.crc32c(bytes(32)) == 0x8A9136AA
    assert S.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert S.crc32c(bytes(range(32))) == 0x46DD794E
    c = S.crc32c(b"abc")
    assert S.masked_crc32c(b"abc") == (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def test_event_file_round_trip_and_histogram_buckets(tmp_path):
    w = S.FileWriter(str(tmp_path / "train"))
    rng = np.random.default_rng(0)
    x = rng.standard_normal((50, 7)) * 3
    w.add_summary(S.model_summaries([1, 2, 3, 4, 5, 6, 7.5], x, np.zeros(5), np.array([1e-13, 1.0, -2.0, 1e30])), 1200)
    w.add_summary(S.merge([S.scalar("lr", 1e-4)]))
    w.close()
    files = glob.glob(str(tmp_path / "train" / "events.out.tfevents.*"))
    assert files == [w.path]
    ev = S.read_events(w.path)                                   # (checks both checksums of every record)
    assert len(ev) == 3 and ev[0][3] == "brain.Event:2" and ev[0][1] is None
    wall, step, vals, _ = ev[1]
    assert step == 1200 and wall > 1e9
    assert [vals[t] for t in S.LOSS_TAGS] == [1, 2, 3, 4, 5, 6, 7.5]
    h = vals["real_clean"]
    assert h["num"] == x.size and np.isclose(h["sum"], x.sum()) and np.isclose(h["sum_squares"], np.square(x).sum())
    assert h["min"] == x.min() and h["max"] == x.max()
    lim, cnt = h["bucket_limit"], h["bucket"]
    assert len(lim) == len(cnt) and np.all(np.diff(lim) > 0) and cnt.sum() == x.size
    # TensorFlow's rule is upper_bound (previous limit <= value < limit); a run of empty buckets is one entry ending at the run's last limit
    assert np.array_equal(np.bincount(np.searchsorted(lim, x.ravel(), side="right"), minlength=len(lim)), cnt)
    assert not np.any((cnt[:-1] == 0) & (cnt[1:] == 0))
    z = vals["real_noise"]                                        # five zeros: one bucket [0, 1e-12)
    assert z["num"] == 5 and z["bucket"].sum() == 5 and np.isclose(z["bucket_limit"][np.nonzero(z["bucket"])[0][0]], 1e-12)
    g = vals["g_clean"]                                           # tiny, ordinary, negative, beyond 1e20 (the DBL_MAX bucket)
    assert g["num"] == 4 and g["bucket_limit"][-1] == np.finfo(np.float64).max and g["bucket"][-1] == 1
    assert ev[2][1] is None and np.isclose(ev[2][2]["lr"], 1e-4)
    # the default limits: 1e-12 * 1.1^k below 1e20, mirrored, with 0 between
    assert np.isclose(S._LIMITS[len(S._LIMITS) // 2], 0.0) and np.isclose(S._LIMITS[len(S._LIMITS) // 2 + 2] / S._LIMITS[len(S._LIMITS) // 2 + 1], 1.1)


class _Model:
    """records what the loops fetch; writer_for / run_summaries come from the real base class"""
    from rsrgan_amd.gan_rnn import Model as _Base
    disc_updates = gen_updates = 1
    cross_validation = False
    process_group = None

    def __init__(self, save_dir):
        self.save_dir = save_dir
        self.calls = []

    _open_writer = _Base._open_writer
    writer_for = _Base.writer_for
    run_summaries = _Base.run_summaries
    writer = None
    _eval_writer = None

    def d_step(self, x, lab, train=True):
        self.calls.append(("d", train))
        return [1.0], [2.0], [3.0]

    def g_step(self, x, lab, train=True):
        self.calls.append(("g", train))
        return [0.5], [0.25], [0.0], [0.75]

    def forward(self, x):
        return np.asarray(x)[:, :1] * 2

    def _summary_fetch(self, inputs, labels, lengths=None):
        # (the contract of Model._summary_fetch: losses of the D and G fetches, then this rank's shard of the batch and G's output on it)
        return (self.d_step(inputs, labels, train=False), self.g_step(inputs, labels, train=False), np.asarray(inputs), np.asarray(labels),
                self.forward(inputs))


def test_frame_level_loops_write_train_and_eval_events(tmp_path):
    """train_gan_dnn.py:132-134,195-196: one summary per epoch pass, step = epoch * num_batch, training fetches under
    save_dir/train and the cross-validation ones under save_dir/eval"""
    from types import SimpleNamespace
    from rsrgan_amd import run_gan_dnn as R
    m = _Model(str(tmp_path / "exp"))
    m._open_writer(SimpleNamespace())
    FLAGS = SimpleNamespace(num_gpu=1)
    batches = [[np.full((4, 2), i, np.float32), np.zeros((4, 1), np.float32)] for i in range(8)]
    R.train_one_epoch(m, batches, 4, 3, FLAGS, log=lambda *_: None)
    assert m.calls == [("d", True), ("g", True)] * 2 + [("d", False), ("g", False)]
    R.eval_one_epoch(m, batches, 4, 3, FLAGS, log=lambda *_: None)
    tr = S.read_events(glob.glob(str(tmp_path / "exp" / "train" / "events*"))[0])
    cv = S.read_events(glob.glob(str(tmp_path / "exp" / "eval" / "events*"))[0])
    assert [e[1] for e in tr] == [None, 12] and [e[1] for e in cv] == [None, 12]
    assert tr[1][2]["g_loss"] == 0.75 and tr[1][2]["d_rl_loss"] == 1.0 and tr[1][2]["real_clean"]["num"] == 8
    assert tr[1][2]["g_clean"]["max"] == 6.0                      # the last batch drawn (i = 3) through forward()
    # without a save_dir (or with write_summaries=False) nothing is opened
    m2 = _Model(None); m2._open_writer(SimpleNamespace())
    m3 = _Model(str(tmp_path / "x")); m3._open_writer(SimpleNamespace(write_summaries=False))
    assert m2.writer is None and m3.writer is None and not os.path.exists(tmp_path / "x")


class _FrameEngine:
    """the slice of the engine protocol a frame-level trainer's summary fetch touches (recording stand-in, CPU)"""
    ema_enabled = False
    device = "cpu"

    def __init__(self):
        self.calls = []

    def set_scalar(self, k, v):
        pass

    def g_backward(self, x, lab, ln, noise_fake=None, train=True, reuse=False, apply=False):
        import torch
        self.calls.append(("g_backward", train, apply))
        return torch.tensor([0.0, 2.0, 0.5, 2.5])

    def forward_g(self, x, ln):
        import torch
        return torch.as_tensor(np.asarray(x))[:, :, :1] * 2


def test_trainers_fetch_summaries_without_a_collective(tmp_path):
    """Model.run_summaries unpacks (d, g, x, labels, y); RNNTrainer / DNNTrainer must honour that contract and fetch
    through the engine directly (g_step all-gathers the towers' losses: a rank-0-only call would hang the other ranks)."""
    import torch
    from tests.helpers import OracleEngine, args_for, rand_batch, rand_params, small_cfg
    from rsrgan_amd.trainer import DNNTrainer, RNNTrainer
    cfg = small_cfg("lstm")
    g, d = rand_params(cfg, 3)
    eng = OracleEngine(cfg, g, d, 4)
    m = RNNTrainer(None, args_for(cfg, 4, save_dir=str(tmp_path / "rnn")), ["cpu"], engine=eng, max_frames=6)
    m._towers = None                                            # any gather in the fetch would blow up here
    x, lab, ln = rand_batch(cfg, 4, 6, seed=5, ragged=True)
    blob = m.run_summaries(x, lab, ln)
    m.writer.add_summary(blob, 7); m.writer.close()
    ev = S.read_events(m.writer.path)
    assert ev[1][1] == 7 and [ev[1][2][t] for t in S.LOSS_TAGS[:3]] == [0.0, 0.0, 0.0]
    assert ev[1][2]["g_loss"] > 0 and ev[1][2]["real_clean"]["num"] == x.size and ev[1][2]["real_noise"]["num"] == lab.size and ev[1][2]["g_clean"]["num"] == 4 * 6 * cfg.output_dim
    fe = _FrameEngine()
    from types import SimpleNamespace
    a = SimpleNamespace(batch_size=4, input_dim=3, output_dim=1, g_type="dnn", save_dir=str(tmp_path / "dnn"))
    t = DNNTrainer(None, a, ["cpu"], engine=fe)
    xb = np.arange(12, dtype=np.float32).reshape(4, 3)
    blob = t.run_summaries(xb, np.zeros((4, 1), np.float32))
    t.writer.add_summary(blob, 3); t.writer.close()
    ev = S.read_events(t.writer.path)
    assert fe.calls == [("g_backward", False, False)]
    assert ev[1][2]["g_loss"] == 2.5 and ev[1][2]["d_loss"] == 0.0 and ev[1][2]["g_clean"]["max"] == 18.0
    with np.testing.assert_raises(RuntimeError):
        t.d_step(xb, xb)
    del torch
