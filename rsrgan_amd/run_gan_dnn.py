#!/usr/bin/env python
"""Outer loop of the frame-level GAN recipe: scripts/train_gan_dnn.py:27-557 (train_one_epoch, eval_one_epoch, decode, train,
main) and stage 2-3 of run_gan_dnn.sh, with Kaldi scp/ark files instead of TFRecords.

    python -m rsrgan_amd.run_gan_dnn --data_dir data/train --tr_inputs_scp tr/inputs.scp --tr_labels_scp tr/labels.scp \\
        --cv_inputs_scp cv/inputs.scp --cv_labels_scp cv/labels.scp --batch_size 256 --batch_norm true --save_dir exp/x
    python -m rsrgan_amd.run_gan_dnn --decode --test_inputs_scp test/inputs.scp --data_dir data/train --save_dir exp/x

Every D-run and every G-run dequeues its own batch of `batch_size` frames (train_gan_dnn.py:56-83: each sess.run pulls from the
RandomShuffleQueue of io_funcs/tfrecords_io.py:233-251).  The cross-validation twin shares every variable with the training
model (train_gan_dnn.py:419-423); here it is the training model fetched with train=False (no update, no L2 term, batch norm in
inference mode).  Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N -m rsrgan_amd.run_gan_dnn ...`; the
reference feeds every tower the SAME batch (models/gan.py:136), here every rank draws its own (seeded by rank) and the
gradients are averaged -- N times the data per step.  Flag names and defaults are the reference's (train_gan_dnn.py:560-740);
the *_list_file flags (lists of TFRecord files) are replaced by *_inputs_scp / *_labels_scp."""
from __future__ import annotations

import argparse
import datetime
import os
import sys

import numpy as np

from . import dist as rdist
from .gan import GAN
from .io import ArkReader, ArkWriter, FrameBatchReader, splice_feats
from .run_gan_rnn import _cmvn, str2bool

LOSS_NAMES = ("d_rl_loss", "d_fk_loss", "d_loss", "g_adv_loss", "g_mse_loss", "g_l2_loss", "g_loss")


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
    p.add_argument("--batch_size", type=int, default=256)
    p.add_argument("--g_learning_rate", type=float, default=0.0001)
    p.add_argument("--d_learning_rate", type=float, default=0.0001)
    p.add_argument("--min_epoches", type=int, default=15)
    p.add_argument("--max_epoches", type=int, default=20)
    p.add_argument("--decay_factor", type=float, default=0.8)
    p.add_argument("--start_decay_impr", type=float, default=0.01)
    p.add_argument("--end_decay_impr", type=float, default=0.001)
    p.add_argument("--num_threads", type=int, default=24)
    p.add_argument("--save_dir", type=str, default="exp/gan")
    p.add_argument("--init_mse_weight", type=float, default=1.0)
    p.add_argument("--g_type", type=str, default="dnn")
    p.add_argument("--disc_updates", type=int, default=1)
    p.add_argument("--gen_updates", type=int, default=1)
    p.add_argument("--batch_norm", type=str2bool, nargs="?", default="false")
    p.add_argument("--init_disc_noise_std", type=float, default=0.0)
    p.add_argument("--keep_lr", type=int, default=3)
    p.add_argument("--keep_prob", type=float, default=1.0)
    p.add_argument("--l2_scale", type=float, default=0.00001)
    p.add_argument("--num_gpu", type=int, default=1)
    p.add_argument("--apply_cmvn", type=str2bool, nargs="?", default="true", help="normalise with data_dir/train_cmvn.npz "
                   "(make_tfrecords.py:84-87 did this when writing TFRecords)")
    return p


def _reader(FLAGS, inputs_scp, labels_scp, cmvn, shuffle, seed):
    return FrameBatchReader(inputs_scp, labels_scp, FLAGS.batch_size, FLAGS.left_context, FLAGS.right_context, cmvn=cmvn,
                            num_threads=FLAGS.num_threads, shuffle=shuffle, seed=seed)


def _one_epoch(model, batches, num_batch, epoch, FLAGS, train, log, report=True):
    """train_one_epoch (:27-150) / eval_one_epoch (:153-215): int(num_batch / (disc_updates + gen_updates) / num_gpu) rounds of
    disc_updates D-runs then gen_updates G-runs, one fresh batch per run; tower means per run (np.mean), run means per epoch."""
    sums, d_counter, g_counter = np.zeros(7), 0, 0
    rep, d_rep, g_rep = np.zeros(7), 0, 0
    start = datetime.datetime.now()
    steps = model.disc_updates + model.gen_updates
    it = iter(batches)
    for _ in range(int(num_batch / steps / FLAGS.num_gpu)):
        try:
            for _d in range(model.disc_updates):
                x, lab = next(it)
                out = model.d_step(x, lab, train=train)
                v = np.array([np.mean(o) for o in out])
                sums[:3] += v; rep[:3] += v; d_counter += 1; d_rep += 1
            for _g in range(model.gen_updates):
                x, lab = next(it)
                out = model.g_step(x, lab, train=train)
                v = np.array([np.mean(o) for o in out])
                sums[3:] += v; rep[3:] += v; g_counter += 1; g_rep += 1
        except StopIteration:                      # tf.errors.OutOfRangeError: the queue ran dry
            break
        counter = (d_counter + g_counter) * FLAGS.num_gpu
        if train and report and (counter / FLAGS.num_gpu) % 1000 == 0:     # :99-128 (the iteration scripts have no such report)
            dur = (datetime.datetime.now() - start).total_seconds()
            log("Epoch {} (BATCH {}): ".format(epoch, counter) +
                ", ".join("{} = {:.5f}".format(n, r / (d_rep if i < 3 else g_rep)) for i, (n, r) in enumerate(zip(LOSS_NAMES, rep))) +
                ", time = {:.3} min".format(dur / 60.0))
            start = datetime.datetime.now()
            rep[:] = 0.0; d_rep = g_rep = 0
    w = model.writer_for(train) if d_counter + g_counter and hasattr(model, "writer_for") else None
    if w is not None:                              # Save summary (:132-134, :195-196): one more fetch on the last batch drawn
        w.add_summary(model.run_summaries(x, lab), epoch * num_batch)
    d_counter, g_counter = max(d_counter, 1), max(g_counter, 1)
    return tuple(np.concatenate([sums[:3] / d_counter, sums[3:] / g_counter]))


def train_one_epoch(model, batches, tr_num_batch, epoch, FLAGS, log=print):
    return _one_epoch(model, batches, tr_num_batch, epoch, FLAGS, True, log)


def eval_one_epoch(model, batches, cv_num_batch, epoch, FLAGS, log=print):
    return _one_epoch(model, batches, cv_num_batch, epoch, FLAGS, False, log)


def _fmt(losses):
    return ", ".join("{} = {:.5f}".format(n, v) for n, v in zip(LOSS_NAMES, losses))


def train(FLAGS, model_factory=None, log=print, net_overrides=None):
    """train (:373-557) + the batch counting of main (:305-343).  Returns the list of per-epoch CV g_loss."""
    cmvn = _cmvn(FLAGS)
    rank = rdist.rank()
    mk = model_factory or (lambda: GAN(None, FLAGS, ["gpu:%d" % rank], net_overrides=net_overrides))
    tr_model = mk()
    cv_model = tr_model                       # shares every variable (:419-423); fetched with train=False
    if tr_model.load(tr_model.save_dir):
        log("[*] Load SUCCESS")
    else:
        log("[!] Begin a new model.")
    tr_reader = _reader(FLAGS, FLAGS.tr_inputs_scp, FLAGS.tr_labels_scp, cmvn, True, 1234 + rank)
    cv_reader = _reader(FLAGS, FLAGS.cv_inputs_scp, FLAGS.cv_labels_scp, cmvn, True, 4321 + rank)
    batch_file = os.path.join(FLAGS.data_dir or ".", "batch_num.txt")                      # :307-324
    if os.path.isfile(batch_file):
        with open(batch_file) as fr:
            cv_num_batch, tr_num_batch = (int(v) for v in fr.readline().strip().split()[:2])
        log("[*] batch_num.txt exist, and cv batches is %d, tr batches is %d." % (cv_num_batch, tr_num_batch))
    else:
        cv_num_batch, tr_num_batch = cv_reader.num_batches(), tr_reader.num_batches()
        if rank == 0:
            with open(batch_file, "w") as fw:
                fw.write("%d %d" % (cv_num_batch, tr_num_batch))
    tr_model.g_learning_rate = FLAGS.g_learning_rate
    tr_model.d_learning_rate = FLAGS.d_learning_rate
    cv = eval_one_epoch(cv_model, cv_reader, cv_num_batch, 0, FLAGS, log)
    log("CROSSVAL.LOSS PRERUN: " + _fmt(cv))
    g_loss_prev, decay_steps, history = cv[6], 1, []
    for epoch in range(FLAGS.max_epoches):
        start = datetime.datetime.now()
        tr = train_one_epoch(tr_model, tr_reader, tr_num_batch, epoch + 1, FLAGS, log)
        if hasattr(tr_model, "sync_batch_norm_state"):
            tr_model.sync_batch_norm_state()           # one copy of the batch-norm statistics over the ranks, like the towers' shared variables
        cv = eval_one_epoch(cv_model, cv_reader, cv_num_batch, epoch + 1, FLAGS, log)
        end = datetime.datetime.now()
        log("Epoch {} (TRAIN AVG.LOSS): {}, d_lr = {:.3e}, g_lr = {:.3e}\nEpoch {} (CROSS AVG.LOSS): {}, time = {:.2f} h".format(
            epoch + 1, _fmt(tr), tr_model.d_learning_rate, tr_model.g_learning_rate, epoch + 1, _fmt(cv), (end - start).seconds / 3600.0))
        g_loss_new = cv[6]
        history.append(g_loss_new)
        # Accept or reject new parameters (:487-507)
        if g_loss_new < g_loss_prev:
            tr_model.save(tr_model.save_dir, epoch + 1)
            log("Epoch {}: Nnet Accepted. Save model SUCCESS.".format(epoch + 1))
            g_rel_impr = (g_loss_prev - g_loss_new) / g_loss_prev
            g_loss_prev = g_loss_new
        else:
            log("Epoch {}: Nnet Rejected.".format(epoch + 1))
            if tr_model.load(tr_model.save_dir):
                log("[*] Load previous model SUCCESS.")
            else:
                log("[!] Load failed. No checkpoint from {} to restore previous model. Exit now.".format(tr_model.save_dir))
                raise SystemExit(1)
            g_rel_impr = (g_loss_prev - g_loss_new) / g_loss_prev
        # Start decay when improvement is low (:510-527)
        if g_rel_impr < FLAGS.start_decay_impr and epoch + 1 >= FLAGS.keep_lr:
            tr_model.g_learning_rate = FLAGS.g_learning_rate * FLAGS.decay_factor ** decay_steps
            tr_model.d_learning_rate = FLAGS.d_learning_rate * FLAGS.decay_factor ** decay_steps
            tr_model.disc_noise_std = FLAGS.init_disc_noise_std * FLAGS.decay_factor ** decay_steps      # (discriminator_dnn's noise layer is commented out, :58)
            decay_steps += 1
        # Stopping criterion (:530-539)
        if g_rel_impr < FLAGS.end_decay_impr:
            if epoch < FLAGS.min_epoches:
                log("Epoch %d: We were supposed to finish, but we continue as min_epoches %d" % (epoch + 1, FLAGS.min_epoches))
                continue
            log("Epoch %d: Finished, too small relative G improvement %g" % (epoch + 1, g_rel_impr))
            break
    log("Training Done.")
    return history


def decode(FLAGS, model_factory=None, log=print, net_overrides=None, moving_average=True):
    """decode (:218-302): the cross_validation graph on one utterance at a time, the exponential moving averages of the
    trainable variables (model.load(moving_average=True), :253), de-normalised with the label CMVN, feats.ark / feats.scp."""
    cmvn = _cmvn(FLAGS)
    reader = ArkReader()
    reader(FLAGS.test_inputs_scp)
    longest = max(reader.utt_shape_from_index(i)[0] for i in range(len(reader.utt_ids)))
    mk = model_factory or (lambda n: GAN(None, argparse.Namespace(**dict(vars(FLAGS), batch_size=n)), ["gpu:%d" % rdist.rank()],
                                         cross_validation=True, net_overrides=net_overrides))
    model = mk(longest)                       # one utterance = one batch of frames (frames are independent in the inference graph)
    if model.load(model.save_dir, moving_average=moving_average):
        log("[*] Load SUCCESS")
    else:
        raise SystemExit("[!] Load failed. Checkpoint not found. Exit now.")
    out_dir = os.path.join(FLAGS.save_dir, "test")
    os.makedirs(out_dir, exist_ok=True)
    write_scp_path, write_ark_path = os.path.join(out_dir, "feats.scp"), os.path.join(out_dir, "feats.ark")
    if os.path.exists(write_ark_path):
        os.remove(write_ark_path)
    writer = ArkWriter(write_scp_path)
    for i, utt in enumerate(reader.utt_ids):
        x = reader.read_utt_data_from_index(i).astype(np.float64)
        if cmvn is not None:
            x = (x - cmvn["mean_inputs"]) / cmvn["stddev_inputs"]
        x = splice_feats(x, FLAGS.left_context, FLAGS.right_context).astype(np.float32)
        n = x.shape[0]
        pad = np.concatenate([x, np.zeros((longest - n, x.shape[1]), np.float32)]) if n < longest else x
        activations = np.asarray(model.forward(pad))[:n]
        sequence = activations * cmvn["stddev_labels"] + cmvn["mean_labels"] if cmvn is not None else activations
        writer.write_next_utt(write_ark_path, utt, np.vstack(sequence))
        log("Write inferred %s to %s" % (utt, write_ark_path))
    writer.close()
    log("Decoding Done.")
    return write_scp_path


def main(argv=None):
    FLAGS, unparsed = build_parser().parse_known_args(argv)
    rank, local, world = rdist.init_from_env()
    if world > 1:
        FLAGS.num_gpu = world
    if FLAGS.decode:
        rdist.run_on_rank0(lambda: decode(FLAGS))                     # one writer; a failure on rank 0 releases (and fails) the others
    else:
        train(FLAGS)


if __name__ == "__main__":
    main()
