#!/usr/bin/env python
"""Outer loop of the sequence-level GAN recipe: scripts/train_gan_rnn_placeholder.py:204-584 (decode, train,
main) and stage 2-3 of run_gan_rnn_placeholder.sh, with Kaldi scp/ark files instead of TFRecords.

    python -m rsrgan_amd.run_gan_rnn --data_dir data/train --tr_inputs_scp tr/inputs.scp --tr_labels_scp tr/labels.scp \\
        --cv_inputs_scp cv/inputs.scp --cv_labels_scp cv/labels.scp --g_type res_lstm_l --batch_size 8 --save_dir exp/x
    python -m rsrgan_amd.run_gan_rnn --decode --test_inputs_scp test/inputs.scp --data_dir data/train --save_dir exp/x

Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N -m rsrgan_amd.run_gan_rnn ...`; each batch of
batch_size*num_gpu utterances is sliced per rank (models/gan_rnn_placeholder.py:157-159), LR x num_gpu (:458-459).
Flag names and defaults are the reference's (train_gan_rnn_placeholder.py:587-747); the *_list_file flags (lists of
TFRecord files) are replaced by *_inputs_scp / *_labels_scp."""
from __future__ import annotations

import argparse
import datetime
import os
import sys

import numpy as np

from . import dist as rdist
from .gan_rnn import GAN_RNN
from .io import ArkReader, ArkWriter, PaddedBatchReader, prefetch, splice_feats
from .train import eval_one_iteration, exponential_decay, train_one_iteration


def str2bool(v):
    return str(v).lower() in ("yes", "true", "t", "1")


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--decode", default=False, action="store_true", help="Flag indicating decoding or training.")
    p.add_argument("--data_dir", type=str, default=None, help="Data directory (holds train_cmvn.npz).")
    for n in ("tr_inputs_scp", "tr_labels_scp", "cv_inputs_scp", "cv_labels_scp", "test_inputs_scp"):
        p.add_argument("--" + n, type=str, default=None)
    p.add_argument("--input_dim", type=int, default=257)
    p.add_argument("--output_dim", type=int, default=40)
    p.add_argument("--left_context", type=int, default=5)
    p.add_argument("--right_context", type=int, default=5)
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--g_learning_rate", type=float, default=0.0003)
    p.add_argument("--d_learning_rate", type=float, default=0.001)
    p.add_argument("--min_epoches", type=int, default=25)
    p.add_argument("--max_epoches", type=int, default=30)
    p.add_argument("--end_improve", type=float, default=0.001)
    p.add_argument("--num_threads", type=int, default=24)
    p.add_argument("--save_dir", type=str, default="exp/gan_rnn")
    p.add_argument("--init_mse_weight", type=float, default=1.0)
    p.add_argument("--g_type", type=str, default="lstm")
    p.add_argument("--disc_updates", type=int, default=1)
    p.add_argument("--gen_updates", type=int, default=2)
    p.add_argument("--batch_norm", type=str2bool, nargs="?", default="false")
    p.add_argument("--keep_prob", type=float, default=1.0)
    p.add_argument("--init_disc_noise_std", type=float, default=0.0)
    p.add_argument("--l2_scale", type=float, default=0.00001)
    p.add_argument("--num_gpu", type=int, default=1)
    p.add_argument("--apply_cmvn", type=str2bool, nargs="?", default="true", help="normalise with data_dir/train_cmvn.npz "
                   "(make_tfrecords.py:84-87 did this when writing TFRecords)")
    p.add_argument("--max_frames", type=int, default=3000, help="capacity of the padded time axis")
    return p


def _cmvn(FLAGS):
    if not str2bool(FLAGS.apply_cmvn):
        return None
    path = os.path.join(FLAGS.data_dir or ".", "train_cmvn.npz")
    if not os.path.isfile(path):
        raise SystemExit("%s not exist, exit now." % path)                 # train_gan_rnn_placeholder.py:256-260
    return np.load(path)


def _reader(FLAGS, inputs_scp, labels_scp, cmvn, shuffle, seed):
    return PaddedBatchReader(inputs_scp, labels_scp, FLAGS.batch_size * FLAGS.num_gpu, FLAGS.left_context, FLAGS.right_context,
                             cmvn=cmvn, shuffle=shuffle, seed=seed)


def get_num_batch(reader, full):
    """get_num_batch (:346-385) counts the batches of a full pass; the reader's plan() gives the same count from the archive
    headers alone (no utterance is read, normalised, spliced or padded just to be counted)."""
    return sum(1 for _ in reader.plan())


def train(FLAGS, model_factory=None, log=print, net_overrides=None):
    """train (:388-584) + the batch counting of main (:305-343).  Returns the list of per-iteration CV g_loss."""
    cmvn = _cmvn(FLAGS)
    mk = model_factory or (lambda cv, share: GAN_RNN(None, FLAGS, ["gpu:%d" % rdist.rank()], cross_validation=cv,
                                                      max_frames=FLAGS.max_frames, share_engine_from=share,
                                                      net_overrides=net_overrides))
    tr_model = mk(False, None)
    cv_model = mk(True, tr_model)                                          # shares variables (:436-437)
    if tr_model.load(tr_model.save_dir, moving_average=False):
        log("[*] Load SUCCESS")
    else:
        log("[!] Begin a new model.")
    full = FLAGS.batch_size * FLAGS.num_gpu
    tr_reader = _reader(FLAGS, FLAGS.tr_inputs_scp, FLAGS.tr_labels_scp, cmvn, True, 1234)
    cv_reader = _reader(FLAGS, FLAGS.cv_inputs_scp, FLAGS.cv_labels_scp, cmvn, False, None)
    tr_num_batch = get_num_batch(_reader(FLAGS, FLAGS.tr_inputs_scp, FLAGS.tr_labels_scp, None, False, None), full)
    cv_num_batch = get_num_batch(_reader(FLAGS, FLAGS.cv_inputs_scp, FLAGS.cv_labels_scp, None, False, None), full)
    train_batch_per_iter, valdi_batch_per_iter = tr_num_batch, cv_num_batch
    min_iters = int(FLAGS.min_epoches * tr_num_batch / train_batch_per_iter)
    max_iters = int(FLAGS.max_epoches * tr_num_batch / train_batch_per_iter)
    log("LOG: #train_batch = {}, #valid_batch = {}, #min_iters = {}, #max_iters = {}".format(tr_num_batch, cv_num_batch, min_iters, max_iters))

    g_loss_prev, g_rel_impr, check_interval, windows_g_loss = 10000.0, 1.0, 1, []           # :452-456
    tr_model.g_learning_rate = FLAGS.num_gpu * FLAGS.g_learning_rate                         # :458-461
    tr_model.d_learning_rate = FLAGS.num_gpu * FLAGS.d_learning_rate
    history = []
    iteration = -1
    for iteration in range(max_iters):
        start = datetime.datetime.now()
        # reader thread + Queue(32) as :463-478: reading, CMVN, splicing and padding overlap the GPU steps
        tr = train_one_iteration(None, tr_model, train_batch_per_iter * FLAGS.num_gpu, iteration + 1, prefetch(tr_reader), FLAGS.num_gpu)
        cv = eval_one_iteration(None, cv_model, valdi_batch_per_iter * FLAGS.num_gpu, iteration + 1,
                                prefetch(b for b in cv_reader if len(b[0]) == full), FLAGS.num_gpu)
        end = datetime.datetime.now()
        log("{}/{} (INFO): d_learning_rate = {:.5e}, g_learning_rate = {:.5e}, time = {:.3f} h\n"
            "{}/{} (TRAIN AVG.LOSS): d_rl_loss = {:.5f}, d_fk_loss = {:.5f}, d_loss = {:.5f}, g_adv_loss = {:.5f}, "
            "g_mse_loss = {:.5f}, g_l2_loss = {:.3e}, g_loss = {:.5f}\n"
            "{}/{} (CROSS AVG.LOSS): d_rl_loss = {:.5f}, d_fk_loss = {:.5f}, d_loss = {:.5f}, g_adv_loss = {:.5f}, "
            "g_mse_loss = {:.5f}, g_l2_loss = {:.3e}, g_loss = {:.5f}".format(
                iteration + 1, max_iters, tr_model.d_learning_rate, tr_model.g_learning_rate, (end - start).total_seconds() / 3600.0,
                iteration + 1, max_iters, *tr, iteration + 1, max_iters, *cv))
        cv_g_loss = cv[6]
        history.append(cv_g_loss)
        # Start decay learning rate (:525-533)
        tr_model.g_learning_rate = exponential_decay(iteration + 1, FLAGS.num_gpu, min_iters, FLAGS.g_learning_rate)
        tr_model.d_learning_rate = exponential_decay(iteration + 1, FLAGS.num_gpu, min_iters, FLAGS.d_learning_rate)
        tr_model.disc_noise_std = exponential_decay(iteration + 1, FLAGS.num_gpu, min_iters, FLAGS.init_disc_noise_std,
                                                    multiply_jobs=False)
        windows_g_loss.append(cv_g_loss)
        # Accept or reject new parameters (:537-554)
        if (iteration + 1) % check_interval == 0:
            g_loss_new = float(np.mean(windows_g_loss))
            g_rel_impr = (g_loss_prev - g_loss_new) / g_loss_prev
            if g_rel_impr > 0.0:
                tr_model.save(tr_model.save_dir, iteration + 1)
                log("Iteration {}: Nnet Accepted. Save model SUCCESS. g_loss_prev = {:.5f}, g_loss_new = {:.5f}".format(
                    iteration + 1, g_loss_prev, g_loss_new))
                g_loss_prev = g_loss_new
            else:
                log("Iteration {}: Nnet Rejected. g_loss_prev = {:.5f}, g_loss_new = {:.5f}".format(iteration + 1, g_loss_prev, g_loss_new))
            windows_g_loss = []
        # Stopping criterion (:557-562)
        if iteration + 1 > min_iters and (iteration + 1) % check_interval == 0 and g_rel_impr < FLAGS.end_improve:
            log("Iteration %d: Finished, too small relative G improvement %g" % (iteration + 1, g_rel_impr))
            break
    if windows_g_loss:                                                     # :570-581
        g_loss_new = float(np.mean(windows_g_loss))
        if (g_loss_prev - g_loss_new) / g_loss_prev > 0.0:
            tr_model.save(tr_model.save_dir, iteration + 1)
    log("Training Done.")
    return history


def decode(FLAGS, model_factory=None, log=print, net_overrides=None):
    """decode (:204-302): batch 1, G(x), de-normalise with the label CMVN, write feats.ark / feats.scp."""
    mk = model_factory or (lambda: GAN_RNN(None, argparse.Namespace(**dict(vars(FLAGS), batch_size=1)), ["gpu:%d" % rdist.rank()],
                                           cross_validation=True, infer=True, max_frames=FLAGS.max_frames,
                                           net_overrides=net_overrides))
    model = mk()
    if model.load(model.save_dir, moving_average=False):
        log("[*] Load SUCCESS")
    else:
        raise SystemExit("[!] Load failed. Checkpoint not found. Exit now.")
    cmvn = _cmvn(FLAGS)
    out_dir = os.path.join(FLAGS.save_dir, "test")
    os.makedirs(out_dir, exist_ok=True)
    write_scp_path, write_ark_path = os.path.join(out_dir, "feats.scp"), os.path.join(out_dir, "feats.ark")
    if os.path.exists(write_ark_path):
        os.remove(write_ark_path)
    writer = ArkWriter(write_scp_path)
    reader = ArkReader()
    reader(FLAGS.test_inputs_scp)
    start = datetime.datetime.now()
    for i, utt in enumerate(reader.utt_ids):
        x = reader.read_utt_data_from_index(i).astype(np.float64)
        if cmvn is not None:
            x = (x - cmvn["mean_inputs"]) / cmvn["stddev_inputs"]
        x = splice_feats(x, FLAGS.left_context, FLAGS.right_context).astype(np.float32)[None]
        activations = np.asarray(model.forward(x, np.array([x.shape[1]], np.int32)))
        sequence = activations * cmvn["stddev_labels"] + cmvn["mean_labels"] if cmvn is not None else activations
        writer.write_next_utt(write_ark_path, utt, np.vstack(sequence))
        log("[{}/{}] Write inferred {} to {}".format(i + 1, len(reader.utt_ids), utt, write_ark_path))
    writer.close()
    log("Decoding time is {}s".format((datetime.datetime.now() - start).total_seconds()))
    return write_scp_path


def main(argv=None):
    FLAGS, unparsed = build_parser().parse_known_args(argv)                # unknown flags are ignored, as in the reference (:748)
    rank, local, world = rdist.init_from_env()
    if world > 1:
        FLAGS.num_gpu = world
    if FLAGS.decode:
        rdist.run_on_rank0(lambda: decode(FLAGS))                          # one writer for <save_dir>/test/feats.{ark,scp}; a failure releases the others
    else:
        train(FLAGS)


if __name__ == "__main__":
    main()
