"""Host logic of the frame-level recipe (scripts/train_gan_dnn.py, io_funcs/tfrecords_io.py:206-255): the random-shuffle frame
reader and the epoch loop's run/batch accounting, on CPU with a recording stand-in for the model."""
import os

import numpy as np
import pytest

from rsrgan_amd import run_gan_dnn as R
from rsrgan_amd.io import ArkWriter, FrameBatchReader


def _data(tmp, n, rng, din=3, dout=2, tag="tr"):
    wi, wl = ArkWriter(str(tmp / (tag + "_in.scp"))), ArkWriter(str(tmp / (tag + "_lab.scp")))
    total = 0
    for i in range(n):
        T = int(rng.integers(40, 90)); total += T
        x = np.zeros((T, din), np.float32); x[:, 0] = i; x[:, 1] = np.arange(T)
        wi.write_next_utt(str(tmp / (tag + "_in.ark")), "%s%02d" % (tag, i), x)
        wl.write_next_utt(str(tmp / (tag + "_lab.ark")), "%s%02d" % (tag, i), 2.0 * x[:, :dout])
    wi.close(); wl.close()
    return str(tmp / (tag + "_in.scp")), str(tmp / (tag + "_lab.scp")), total


def test_frame_reader_draws_every_frame_once_and_pairs_labels(tmp_path):
    rng = np.random.default_rng(0)
    xs, ls, total = _data(tmp_path, 25, rng)
    r = FrameBatchReader(xs, ls, 32, left_context=1, right_context=1, num_threads=2, seed=3)
    assert r.capacity == 1000 + 3 * 32 and r.num_batches() == total // 32
    batches = list(r)
    assert len(batches) == total // 32                       # the frames that do not fill a batch are dropped
    seen = set()
    for x, y in batches:
        assert x.shape == (32, 9) and y.shape == (32, 2) and x.dtype == np.float32
        assert np.array_equal(y, 2.0 * x[:, 3:5])            # centre frame of the spliced input <-> its label
        seen.update((int(a), int(b)) for a, b in zip(x[:, 3], x[:, 4]))
    assert len(seen) == 32 * len(batches)                    # no frame twice
    # shuffled: the first batch mixes utterances; a second pass differs from the first, the same seed repeats it
    assert len({int(v) for v in batches[0][0][:, 3]}) > 3
    again = list(FrameBatchReader(xs, ls, 32, 1, 1, num_threads=2, seed=3))
    assert all(np.array_equal(a[0], b[0]) for a, b in zip(batches, again))
    # splice context: the left neighbour of frame t is frame t-1 of the same utterance (first frame repeated)
    x = batches[0][0]
    assert np.all(x[:, 0] == x[:, 3]) and np.all(x[:, 1] == np.maximum(x[:, 4] - 1, 0))
    plain = list(FrameBatchReader(xs, ls, 32, 0, 0, shuffle=False))
    assert np.array_equal(plain[0][0][:3, :2], [[0, 0], [0, 1], [0, 2]])


class _Recorder:
    disc_updates, gen_updates = 1, 2

    def __init__(self):
        self.calls = []

    def d_step(self, x, lab, train=True):
        self.calls.append(("d", float(x[0, 0]), train))
        return [1.0, 3.0], [2.0, 2.0], [3.0, 5.0]

    def g_step(self, x, lab, train=True):
        self.calls.append(("g", float(x[0, 0]), train))
        return [0.5], [0.25], [0.0], [0.75]


def test_epoch_loop_runs_and_batches():
    FLAGS, _ = R.build_parser().parse_known_args(["--num_gpu", "1"])
    assert (FLAGS.batch_size, FLAGS.min_epoches, FLAGS.max_epoches, FLAGS.decay_factor, FLAGS.keep_lr) == (256, 15, 20, 0.8, 3)
    m = _Recorder()
    batches = ([np.full((4, 2), i, np.float32), np.zeros((4, 1), np.float32)] for i in range(100))
    out = R.train_one_epoch(m, batches, 10, 1, FLAGS, log=lambda *_: None)
    # int(10 / (1 + 2) / 1) = 3 rounds of 1 D-run + 2 G-runs, a fresh batch per run (train_gan_dnn.py:52-83)
    assert [c[0] for c in m.calls] == ["d", "g", "g"] * 3
    assert [c[1] for c in m.calls] == list(range(9)) and all(c[2] for c in m.calls)
    assert np.allclose(out, (2.0, 2.0, 4.0, 0.5, 0.25, 0.0, 0.75))          # tower means per run, run means per epoch
    m.calls.clear()
    R.eval_one_epoch(m, batches, 7, 1, FLAGS, log=lambda *_: None)
    assert len(m.calls) == 6 and not any(c[2] for c in m.calls)
    m.calls.clear()
    short = ([np.zeros((4, 2), np.float32), np.zeros((4, 1), np.float32)] for _ in range(4))
    R.train_one_epoch(m, short, 30, 1, FLAGS, log=lambda *_: None)         # the queue runs dry: OutOfRangeError ends the epoch
    assert len(m.calls) == 4


@pytest.mark.timeout(60)
def test_frame_reader_admits_an_utterance_longer_than_the_queue(tmp_path):
    """An utterance that exceeds capacity - residual frames must not stall the reader (tf's enqueue_many admits it piecewise):
    lengths [300, 1400, 200] at batch 32 hung the first version forever."""
    wi, wl = ArkWriter(str(tmp_path / "in.scp")), ArkWriter(str(tmp_path / "lab.scp"))
    lens = [300, 1400, 200]
    for i, T in enumerate(lens):
        x = np.zeros((T, 2), np.float32); x[:, 0] = i; x[:, 1] = np.arange(T)
        wi.write_next_utt(str(tmp_path / "in.ark"), "u%d" % i, x)
        wl.write_next_utt(str(tmp_path / "lab.ark"), "u%d" % i, x[:, :1])
    wi.close(); wl.close()
    for shuffle in (False, True):
        r = FrameBatchReader(str(tmp_path / "in.scp"), str(tmp_path / "lab.scp"), 32, num_threads=1, shuffle=shuffle, seed=1)
        assert r.capacity == 1000 + 2 * 32 < 1400
        it, batches = iter(r), []
        for _ in range(sum(lens) // 32 + 2):                   # bounded: a stalled iterator fails instead of hanging the suite
            b = next(it, None)
            if b is None:
                break
            batches.append(b)
        assert len(batches) == sum(lens) // 32
        seen = {(int(a), int(b)) for x, _ in batches for a, b in x}
        assert len(seen) == 32 * len(batches)


class _Scripted(_Recorder):
    """a model whose cross-validation g_loss follows a script (one value per eval_one_iteration)"""
    disc_updates, gen_updates = 1, 1

    def __init__(self, cv_g_loss, save_dir):
        super().__init__()
        self.script, self.k, self.saved, self.save_dir = list(cv_g_loss), -1, [], save_dir
        self.lrs = []

    def load(self, save_dir, moving_average=False):
        return False

    def save(self, save_dir, step):
        self.saved.append(step)

    def d_step(self, x, lab, train=True):
        if not train and not self._in_eval:
            self.k += 1; self._in_eval = True
        if train:
            self._in_eval = False
            self.lrs.append((self.d_learning_rate, self.g_learning_rate))
        return super().d_step(x, lab, train)

    _in_eval = False

    def g_step(self, x, lab, train=True):
        super().g_step(x, lab, train)
        return [0.5], [0.25], [0.0], [0.75 if train else self.script[min(self.k, len(self.script) - 1)]]


def test_iteration_schedule_matches_the_reference_rules(tmp_path):
    """scripts/train_gan_dnn_iter.py: iteration sizes (:273-290), exponential_decay after every iteration from num_gpu * lr
    (:414-419, :462-474), windows of three CV g_losses, accept -> checkpoint / reject -> nothing restored (:479-498), stop after
    min_iters on a checked window below end_improve (:500-506), the incomplete last window (:509-521)."""
    from rsrgan_amd import run_gan_dnn_iter as RI
    from rsrgan_amd.train import exponential_decay
    rng = np.random.default_rng(0)
    tr = _data(tmp_path, 6, rng, tag="tr"); cv = _data(tmp_path, 3, rng, tag="cv")
    FLAGS, _ = RI.build_parser().parse_known_args([
        "--data_dir", str(tmp_path), "--tr_inputs_scp", tr[0], "--tr_labels_scp", tr[1], "--cv_inputs_scp", cv[0], "--cv_labels_scp", cv[1],
        "--input_dim", "3", "--output_dim", "2", "--left_context", "0", "--right_context", "0", "--batch_size", "16", "--apply_cmvn", "false",
        "--min_epoches", "1", "--max_epoches", "3", "--num_threads", "1", "--g_learning_rate", "0.01", "--d_learning_rate", "0.02",
        "--save_dir", str(tmp_path / "exp")])
    assert FLAGS.end_improve == 0.001 and not hasattr(FLAGS, "decay_factor")
    # sizes: 15000 * 256 / 16 > tr batches -> one iteration = one epoch's worth; with 40 train batches and an iteration of 10:
    FLAGS.batch_size = 256 * 1500                                  # 15000 * 256 / batch = 10, 2000 * 256 / batch = 1.33
    v, t, mn, mx = RI.schedule(FLAGS, cv_num_batch=5, tr_num_batch=40)
    assert (t, mn, mx) == (10.0, 4, 12) and abs(v - 2000 * 256 / FLAGS.batch_size) < 1e-12
    FLAGS.batch_size = 16
    # a scripted run: 4 train batches per iteration (2 rounds of 1 D + 1 G), max_iters = int(3 * 4 / 4) = 3 ... use explicit counts
    script = [5.0, 4.0, 3.0,      # window 1: mean 4.0  < 10000 -> accepted at iteration 3
              4.5, 4.5, 4.5,      # window 2: mean 4.5  > 4.0   -> rejected at 6 (nothing restored), rel impr < 0 -> stop needs it > min_iters
              1.0]
    m = _Scripted(script, str(tmp_path / "exp"))
    logs = []
    FLAGS.min_epoches, FLAGS.max_epoches = 4, 7                     # -> min_iters 4, max_iters 7 with one iteration = "one epoch"
    hist = RI.train(FLAGS, model_factory=lambda: m, log=logs.append, batch_counts=(2, 4))
    text = "\n".join(logs)
    assert "#min_iters = 4, #max_iters = 7" in text
    assert hist == script[:6]                                     # stopped at iteration 6 (> min_iters, checked window, rel impr < end_improve)
    assert m.saved == [3]
    assert "Iteration 3: Nnet Accepted" in text and "Iteration 6: Nnet Rejected" in text and "Iteration 6: Finished" in text
    # learning rates seen by the training runs of iteration k+1 = exponential_decay(k, 1, min_iters, lr); the first = num_gpu * lr
    per_iter = [m.lrs[i] for i in range(0, len(m.lrs), 2)]         # 2 D-runs per iteration
    assert per_iter[0] == (0.02, 0.01)
    for k in range(1, 6):
        assert np.allclose(per_iter[k], (exponential_decay(k, 1, 4, 0.02), exponential_decay(k, 1, 4, 0.01)), rtol=1e-12)
    assert np.isclose(per_iter[4][1], 0.01 * 1e-4) and np.isclose(per_iter[5][1], 0.01 * 1e-4)       # final rate from iteration min_iters on
    # an incomplete last window is still judged: max_iters = 4 -> iterations 1-3 checked, iteration 4 alone at the end
    m2 = _Scripted([5.0, 4.0, 3.0, 1.0], str(tmp_path / "exp"))
    FLAGS.min_epoches, FLAGS.max_epoches = 4, 4
    assert RI.train(FLAGS, model_factory=lambda: m2, log=logs.append, batch_counts=(2, 4)) == [5.0, 4.0, 3.0, 1.0]
    assert m2.saved == [3, 4]
    # the training queue runs on across iterations: batches are not replayed from the start of the data
    firsts = [c[1] for c in m2.calls if c[0] == "d" and c[2]]
    assert len(firsts) == 8
