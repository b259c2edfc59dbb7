"""Host logic of the frame-level recipe (scripts/train_gan_dnn.py, io_funcs/tfrecords_io.py:206-255): the random-shuffle frame
reader and the epoch loop's run/batch accounting, on CPU with a recording stand-in for the model."""
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
