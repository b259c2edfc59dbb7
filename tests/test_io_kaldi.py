"""NumPy/Kaldi-ark data path (rsrgan_amd/io) against archives decoded by the REFERENCE's own
io_funcs/kaldi_io.py:ArkReader (fixture tests/golden/ark_golden.npz, made by make_ark_golden.py)."""
import os
import struct

import numpy as np
import pytest

from rsrgan_amd.io import ArkReader, ArkWriter, PaddedBatchReader, apply_cmvn, convert_cmvn_to_numpy, read_binary_file, splice_feats

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ark_golden.npz"))
UTTS = ["utt_f32", "utt_f64", "utt_cmp", "utt_one"]


@pytest.fixture()
def ark_dir(tmp_path):
    ark = tmp_path / "feats.ark"
    ark.write_bytes(GOLD["ark_bytes"].tobytes())
    scp = tmp_path / "feats.scp"
    scp.write_text("".join(str(l).replace("feats.ark", str(ark)) + "\n" for l in GOLD["scp_lines"]))
    return tmp_path, str(ark), str(scp)


def test_reader_is_bit_identical_to_reference_reader(ark_dir):
    _, ark, scp = ark_dir
    r = ArkReader()
    r(scp)
    assert r.utt_ids == UTTS
    for utt in UTTS:
        uid, data, looped = r.read_next_utt()
        ref = GOLD["ref/" + utt]
        assert uid == utt and not looped
        assert data.dtype == ref.dtype and data.shape == ref.shape
        assert np.array_equal(data, ref), utt                      # float, double AND compressed: bit-exact
    uid, _, looped = r.read_next_utt()
    assert looped and uid == UTTS[0]                               # loops around (kaldi_io.py:174-176)
    assert np.array_equal(r.read_utt_data_from_id("utt_cmp"), GOLD["ref/utt_cmp"])
    # compressed decode stays within the quantisation step of the source matrix
    src = GOLD["src/utt_cmp"]
    assert np.abs(GOLD["ref/utt_cmp"] - src).max() < (src.max() - src.min()) / 60.0
    r.scp_position = 0
    assert [r.read_next_scp() for _ in range(5)] == UTTS + UTTS[:1]
    r.scp_position = 2
    r.split()
    assert r.utt_ids == UTTS[2:-1]                                 # the reference drops the last entry too


def test_reader_errors(tmp_path):
    bad = tmp_path / "bad.ark"
    bad.write_bytes(b"u\x00XFM " + b"\x00" * 20)
    with pytest.raises(ValueError, match="not binary"):
        ArkReader().read_ark(str(bad), 1)
    emp = tmp_path / "emp.ark"
    emp.write_bytes(b"u\x00BCM " + struct.pack("<ffii", 0.0, 1.0, 3, 0))
    with pytest.raises(ValueError, match="Empty"):
        ArkReader().read_ark(str(emp), 1)
    r = ArkReader()
    assert r.read_next_utt() == (None, None, True)                 # empty scp (:170-171)


def test_writer_layout_and_roundtrip(tmp_path):
    w = ArkWriter(str(tmp_path / "out.scp"))
    ark = str(tmp_path / "out.ark")
    mats = {"utt_f32": GOLD["src/utt_f32"], "b": np.arange(6, dtype=np.float64).reshape(2, 3)}
    for k, m in mats.items():
        w.write_next_utt(ark, k, m)
    w.close()
    blob = open(ark, "rb").read()
    # same bytes as the archive the reference reader decoded (utt_f32 is its first entry)
    n = len("utt_f32") + 15 + mats["utt_f32"].size * 4
    assert blob[:n] == GOLD["ark_bytes"].tobytes()[:n]
    r = ArkReader()
    r(str(tmp_path / "out.scp"))
    for k, m in mats.items():
        got = r.read_utt_data_from_id(k)
        assert got.dtype == np.float32 and np.array_equal(got, m.astype(np.float32))
    assert np.array_equal(read_binary_file(ark, len("utt_f32")), mats["utt_f32"])


def test_cmvn_conversion_and_application(tmp_path):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((50, 6)) * 2 + 1
    y = rng.standard_normal((50, 3)) - 4
    for name, m in (("in.cmvn", x), ("lab.cmvn", y)):             # Kaldi global stats: [2, D+1] doubles
        stats = np.zeros((2, m.shape[1] + 1))
        stats[0, :-1] = m.sum(0); stats[0, -1] = m.shape[0]; stats[1, :-1] = (m ** 2).sum(0)
        (tmp_path / name).write_bytes(b"\x00BDM " + struct.pack("<bi", 4, 2) + struct.pack("<bi", 4, m.shape[1] + 1) + stats.tobytes())
    cm = convert_cmvn_to_numpy(str(tmp_path / "in.cmvn"), str(tmp_path / "lab.cmvn"), str(tmp_path))
    assert np.allclose(cm["mean_inputs"], x.mean(0)) and np.allclose(cm["stddev_inputs"], x.std(0))
    saved = np.load(tmp_path / "train_cmvn.npz")
    assert set(saved.files) == {"mean_inputs", "stddev_inputs", "mean_labels", "stddev_labels"}
    xn, yn = apply_cmvn(x, y, saved)
    assert np.allclose(xn.mean(0), 0, atol=1e-12) and np.allclose(yn.std(0), 1)


def test_splice_feats_matches_kaldi_edge_replication():
    f = np.arange(12, dtype=np.float32).reshape(4, 3)             # rows r0..r3
    s = splice_feats(f, 2, 1)
    assert s.shape == (4, 12)
    want_rows = [[0, 0, 0, 1], [0, 0, 1, 2], [0, 1, 2, 3], [1, 2, 3, 3]]      # [t-2, t-1, t, t+1] clamped
    for t, rows in enumerate(want_rows):
        assert np.array_equal(s[t], np.concatenate([f[r] for r in rows]))
    assert np.array_equal(splice_feats(f, 0, 0), f)
    one = splice_feats(f[:1], 3, 3)
    assert np.array_equal(one, np.tile(f[:1], 7))


def test_padded_batches_bucket_by_length_and_feed_the_training_loop(tmp_path):
    rng = np.random.default_rng(1)
    lens = [180, 210, 255, 260, 300, 190, 249, 251, 1400, 1401]
    wi, wl = ArkWriter(str(tmp_path / "in.scp")), ArkWriter(str(tmp_path / "lab.scp"))
    for i, n in enumerate(lens):
        wi.write_next_utt(str(tmp_path / "in.ark"), "u%02d" % i, rng.standard_normal((n, 9)))
        wl.write_next_utt(str(tmp_path / "lab.ark"), "u%02d" % i, rng.standard_normal((n, 5)))
    wi.close(); wl.close()
    rd = PaddedBatchReader(str(tmp_path / "in.scp"), str(tmp_path / "lab.scp"), 2, left_context=1, right_context=1, shuffle=False)
    batches = list(rd)
    key = lambda n: min(20, (n - 200) // 50)
    seen = []
    for ids, X, Y, L in batches:
        assert X.dtype == np.float32 and X.shape[2] == 27 and Y.shape[2] == 5 and X.shape[1] == L.max()
        assert len({key(int(n)) for n in L}) == 1                  # one bucket per batch (tfrecords_dataset.py:157-165)
        for b, n in enumerate(L):
            assert np.all(X[b, n:] == 0) and np.all(Y[b, n:] == 0)
        seen += ids
    assert sorted(seen) == ["u%02d" % i for i in range(len(lens))]
    full = [b for b in batches if len(b[0]) == 2]
    assert len(full) == 4 and len(batches) == 6                    # (180,190) (210,249) (255,260) (1400,1401) + partial 300, 251
    # the items are what train_one_iteration pops: [ids, inputs, labels, lengths]
    from oracle import rsrgan_oracle as O
    from rsrgan_amd import GAN_RNN, train_one_iteration
    from tests.helpers import OracleEngine, args_for, rand_params
    cfg = O.NetCfg(input_dim=27, output_dim=5, g_type="lstm", g_layers=1, g_cells=6, g_proj=4, d_layers=1, d_cells=4, d_proj=3)
    g, d = rand_params(cfg, 3)
    m = GAN_RNN(None, args_for(cfg, 2, input_dim=9, left_context=1, right_context=1), ["cpu:0"], engine=OracleEngine(cfg, g, d, 2))
    short = [[ids, X[:, :6], Y[:, :6], np.minimum(L, 6)] for ids, X, Y, L in batches]
    res = train_one_iteration(None, m, len(short), 0, short)
    assert len(res) == 7 and m.engine.o.adam_t == 4                # the two partial batches were skipped (:69-70)


def test_prefetch_keeps_order_propagates_errors_and_stops():
    from rsrgan_amd.io import prefetch
    items = [[["u%d" % i], np.full((1, 2, 3), i, np.float32), np.zeros((1, 2, 1), np.float32), np.array([2], np.int32)] for i in range(50)]
    got = list(prefetch(iter(items), capacity=4))
    assert [g[0] for g in got] == [it[0] for it in items]
    assert all(float(np.asarray(g[1]).ravel()[0]) == i for i, g in enumerate(got))

    def bad():
        yield items[0]
        raise ValueError("boom")
    it = prefetch(bad(), capacity=2)
    assert next(it)[0] == ["u0"]
    with pytest.raises(ValueError):
        next(it)
    # abandoning the consumer must not leave the producer blocked on a full queue
    import threading
    n0 = threading.active_count()
    g = prefetch(iter(items), capacity=1)
    next(g); g.close()
    import time
    for _ in range(50):
        if threading.active_count() <= n0:
            break
        time.sleep(0.05)
    assert threading.active_count() <= n0


def test_threaded_reader_equals_direct_iteration(tmp_path):
    from rsrgan_amd.io import ArkWriter, PaddedBatchReader, prefetch
    rng = np.random.default_rng(3)
    wi, wl = ArkWriter(str(tmp_path / "in.scp")), ArkWriter(str(tmp_path / "lab.scp"))
    for i in range(37):
        t = int(rng.integers(190, 330))
        wi.write_next_utt(str(tmp_path / "in.ark"), "u%03d" % i, rng.standard_normal((t, 6)).astype(np.float32))
        wl.write_next_utt(str(tmp_path / "lab.ark"), "u%03d" % i, rng.standard_normal((t, 3)).astype(np.float32))
    wi.close(); wl.close()
    cmvn = dict(mean_inputs=rng.standard_normal(6), stddev_inputs=1 + rng.random(6), mean_labels=rng.standard_normal(3), stddev_labels=1 + rng.random(3))
    mk = lambda: PaddedBatchReader(str(tmp_path / "in.scp"), str(tmp_path / "lab.scp"), 4, left_context=1, right_context=2, cmvn=cmvn, shuffle=True, seed=5)
    a = list(mk())
    b = list(prefetch(mk(), capacity=3, threads=3))
    assert len(a) == len(b) and len(a) >= 9
    for x, y in zip(a, b):
        assert x[0] == y[0]
        for u, v in zip(x[1:], y[1:]):
            assert np.array_equal(u, v)
    assert mk().inputs.utt_shape_from_index(0) == mk().inputs.read_utt_data_from_index(0).shape
