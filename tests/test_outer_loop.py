"""scripts/train_gan_rnn_placeholder.py outer loop (train: decay, accept/reject, early stop; decode -> ark) on CPU through
the oracle-backed engine and real scp/ark files."""
import argparse
import os

import numpy as np

from oracle import rsrgan_oracle as O
from rsrgan_amd import GAN_RNN
from rsrgan_amd import run_gan_rnn as R
from rsrgan_amd.io import ArkReader, ArkWriter
from tests.helpers import OracleEngine, rand_params


def _data(tmp, n, tag, rng, din, dout):
    wi, wl = ArkWriter(str(tmp / (tag + "_in.scp"))), ArkWriter(str(tmp / (tag + "_lab.scp")))
    for i in range(n):
        T = int(rng.integers(6, 10))
        wi.write_next_utt(str(tmp / (tag + "_in.ark")), "%s%02d" % (tag, i), rng.standard_normal((T, din)) * 2 + 1)
        wl.write_next_utt(str(tmp / (tag + "_lab.ark")), "%s%02d" % (tag, i), rng.standard_normal((T, dout)) - 1)
    wi.close(); wl.close()
    return str(tmp / (tag + "_in.scp")), str(tmp / (tag + "_lab.scp"))


def test_train_then_decode_end_to_end(tmp_path):
    rng = np.random.default_rng(0)
    din, dout = 5, 3
    tr = _data(tmp_path, 8, "tr", rng, din, dout)
    cv = _data(tmp_path, 4, "cv", rng, din, dout)
    te = _data(tmp_path, 3, "te", rng, din, dout)
    np.savez(tmp_path / "train_cmvn.npz", mean_inputs=np.full(din, 1.0), stddev_inputs=np.full(din, 2.0),
             mean_labels=np.full(dout, -1.0), stddev_labels=np.ones(dout))
    FLAGS, _ = R.build_parser().parse_known_args([
        "--data_dir", str(tmp_path), "--tr_inputs_scp", tr[0], "--tr_labels_scp", tr[1], "--cv_inputs_scp", cv[0],
        "--cv_labels_scp", cv[1], "--test_inputs_scp", te[0], "--input_dim", str(din), "--output_dim", str(dout),
        "--left_context", "1", "--right_context", "1", "--batch_size", "2", "--min_epoches", "2", "--max_epoches", "3",
        "--save_dir", str(tmp_path / "exp"), "--g_learning_rate", "0.003", "--init_disc_noise_std", "0.05",
        "--start_halving_impr", "0.01"])                                   # unknown flag ignored like the reference
    cfg = O.NetCfg(input_dim=din * 3, output_dim=dout, g_type="lstm", g_layers=1, g_cells=6, g_proj=4, d_layers=1, d_cells=4, d_proj=3)
    g, d = rand_params(cfg, 5)
    engines = []

    def factory(cv_flag, share):
        if share is not None:
            return GAN_RNN(None, FLAGS, ["cpu:0"], cross_validation=True, share_engine_from=share)
        engines.append(OracleEngine(cfg, g, d, 2, l2_scale=FLAGS.l2_scale))
        return GAN_RNN(None, FLAGS, ["cpu:0"], engine=engines[-1])
    logs = []
    hist = R.train(FLAGS, model_factory=factory, log=logs.append)
    assert 1 <= len(hist) <= 3 and all(np.isfinite(hist))
    text = "\n".join(logs)
    assert "Nnet Accepted" in text and "Training Done." in text
    assert os.path.exists(tmp_path / "exp" / "checkpoint")
    o = engines[0].o
    # after iteration k the LR is exponential_decay(k, num_gpu, min_iters, init) (utils/ops.py:378-391)
    k = len(hist)
    assert np.isclose(o.g_learning_rate, np.float32(O.exponential_decay(k, 1, 2, 0.003)), rtol=1e-6)
    # decode: batch 1, de-normalised, written to <save_dir>/test/feats.{ark,scp}
    FLAGS.decode = True
    eng_dec = OracleEngine(cfg, g, d, 1)
    scp = R.decode(FLAGS, model_factory=lambda: GAN_RNN(None, argparse.Namespace(**dict(vars(FLAGS), batch_size=1)), ["cpu:0"],
                                                         cross_validation=True, infer=True, engine=eng_dec), log=logs.append)
    r = ArkReader()
    r(scp)
    src = ArkReader(); src(te[0])
    assert r.utt_ids == src.utt_ids
    for i in range(len(r.utt_ids)):
        out = r.read_utt_data_from_index(i)
        assert out.shape == (src.read_utt_data_from_index(i).shape[0], dout) and np.all(np.isfinite(out))
    # decoded with the weights of the accepted checkpoint, not the initial ones
    assert any(not np.allclose(eng_dec.o.g[k], g[k]) for k in g)
