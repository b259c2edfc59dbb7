#!/usr/bin/env python
"""Outer loop of the frame-level GAN recipe on the ITERATION schedule: scripts/train_gan_dnn_iter.py:26-530 (run_gan_dnn_iter.sh)
and its tf.data twin scripts/train_gan_dnn_iter_dataset.py (run_gan_dnn_iter_dataset.sh; same schedule, another input pipeline
and `batch_num_frame_<batch_size>.txt` -- select with --dataset), with Kaldi scp/ark files instead of TFRecords.

Differences from the epoch schedule (rsrgan_amd/run_gan_dnn.py = scripts/train_gan_dnn.py):
  * one "iteration" = 15000 * (256 / batch_size) training batches (at most one epoch) drawn from ONE queue that runs on across
    iterations, then 2000 * (256 / batch_size) cross-validation batches (:273-290);
  * min_iters / max_iters = int({min,max}_epoches * tr_num_batch / train_batch_per_iter);
  * learning rates follow utils/ops.py:378-391 exponential_decay(iteration + 1, num_gpu, min_iters, lr) after EVERY iteration,
    starting from num_gpu * lr (:414-419, :462-474);
  * every check_interval = 3 iterations the mean CV g_loss of the window is compared with the last accepted one: accepted ->
    checkpoint; rejected -> nothing is restored (:479-498, the reload is commented out); training stops after min_iters when the
    relative improvement of a checked window is below --end_improve (:500-506);
  * no CROSSVAL PRERUN, no per-1000-batch report, checkpoints are loaded without the moving averages (:400).
"""
from __future__ import annotations

import datetime
import os

import numpy as np

from . import dist as rdist
from .gan import GAN
from .run_gan_dnn import LOSS_NAMES, _cmvn, _fmt, _one_epoch, _reader, build_parser as _epoch_parser, decode
from .train import exponential_decay

CHECK_INTERVAL = 3            # train_gan_dnn_iter.py:411


def build_parser():
    p = _epoch_parser()
    for a in list(p._actions):                                  # the iteration scripts drop these three flags (:560-700)
        if a.dest in ("decay_factor", "start_decay_impr", "end_decay_impr"):
            p._remove_action(a)
            for o in a.option_strings:
                p._option_string_actions.pop(o, None)
    p.add_argument("--end_improve", type=float, default=0.001, help="Stop when relative loss is lower than end_improve.")
    p.add_argument("--dataset", default=False, action="store_true",
                   help="train_gan_dnn_iter_dataset.py: the batch counts live in batch_num_frame_<batch_size>.txt")
    return p


def _stream(reader, epochs):
    """get_batch(..., num_epochs): one queue for the whole run (train: max_epoches passes, cv: None = for ever)"""
    e = 0
    while epochs is None or e < epochs:
        n = 0
        for b in reader:
            n += 1
            yield b
        if n == 0:
            return
        e += 1


def train_one_iteration(model, batches, tr_num_batch, iteration, FLAGS, log=print):
    return _one_epoch(model, batches, tr_num_batch, iteration, FLAGS, True, log, report=False)


def eval_one_iteration(model, batches, cv_num_batch, iteration, FLAGS, log=print):
    return _one_epoch(model, batches, cv_num_batch, iteration, FLAGS, False, log, report=False)


def schedule(FLAGS, cv_num_batch, tr_num_batch):
    """main (:273-290) -> (valid_batch_per_iter, train_batch_per_iter, min_iters, max_iters)"""
    train_batch_per_iter = 15000 * (256 / FLAGS.batch_size)
    valid_batch_per_iter = 2000 * (256 / FLAGS.batch_size)
    train_batch_per_iter = min(train_batch_per_iter, tr_num_batch)
    valid_batch_per_iter = min(valid_batch_per_iter, cv_num_batch)
    min_iters = int(FLAGS.min_epoches * tr_num_batch / train_batch_per_iter)
    max_iters = int(FLAGS.max_epoches * tr_num_batch / train_batch_per_iter)
    return valid_batch_per_iter, train_batch_per_iter, min_iters, max_iters


def train(FLAGS, model_factory=None, log=print, net_overrides=None, batch_counts=None):
    """main's batch counting (:253-290) + train (:343-530).  Returns the list of per-iteration CV g_loss."""
    cmvn = _cmvn(FLAGS)
    rank = rdist.rank()
    mk = model_factory or (lambda: GAN(None, FLAGS, ["gpu:%d" % rank], net_overrides=net_overrides))
    tr_model = mk()
    cv_model = tr_model                       # shares every variable (:384-389); fetched with train=False
    if tr_model.load(tr_model.save_dir):      # moving_average=False (:400)
        log("[*] Load SUCCESS")
    else:
        log("[!] Begin a new model.")
    tr_reader = _reader(FLAGS, FLAGS.tr_inputs_scp, FLAGS.tr_labels_scp, cmvn, True, 1234 + rank)
    cv_reader = _reader(FLAGS, FLAGS.cv_inputs_scp, FLAGS.cv_labels_scp, cmvn, True, 4321 + rank)
    if batch_counts is not None:
        cv_num_batch, tr_num_batch = batch_counts
    else:
        filename = ("batch_num_frame_%s.txt" if FLAGS.dataset else "batch_num_%s.txt") % FLAGS.batch_size          # :256
        batch_file = os.path.join(FLAGS.data_dir or ".", filename)
        if os.path.isfile(batch_file):
            with open(batch_file) as fr:
                cv_num_batch, tr_num_batch = (int(v) for v in fr.readline().strip().split()[:2])
            log("LOG: %s exist, cross validation batches is %d, trian batches is %d." % (filename, cv_num_batch, tr_num_batch))
        else:
            cv_num_batch, tr_num_batch = cv_reader.num_batches(), tr_reader.num_batches()
            if rank == 0:
                with open(batch_file, "w") as fw:
                    fw.write("%d %d" % (cv_num_batch, tr_num_batch))
    valid_batch_per_iter, train_batch_per_iter, min_iters, max_iters = schedule(FLAGS, cv_num_batch, tr_num_batch)
    log("\nLOG: #train_batch = {}, #valid_batch = {}\nLOG: #batch_per_train_iter = {}, #batch_per_valid_iter = {}\n"
        "LOG: #min_epoches = {}, #max_epoches = {}\nLOG: #min_iters = {}, #max_iters = {}, #itres_per_epoch = {:.2f}\n".format(
            tr_num_batch, cv_num_batch, train_batch_per_iter, valid_batch_per_iter, FLAGS.min_epoches, FLAGS.max_epoches, min_iters,
            max_iters, max_iters / FLAGS.max_epoches))
    tr_batches, cv_batches = _stream(tr_reader, FLAGS.max_epoches), _stream(cv_reader, None)
    # Early stop counter (:407-411)
    g_loss_prev, g_rel_impr, windows_g_loss, history = 10000.0, 1.0, [], []
    tr_model.g_learning_rate = FLAGS.num_gpu * FLAGS.g_learning_rate
    tr_model.d_learning_rate = FLAGS.num_gpu * FLAGS.d_learning_rate
    iteration = -1
    for iteration in range(max_iters):
        start = datetime.datetime.now()
        tr = train_one_iteration(tr_model, tr_batches, train_batch_per_iter, iteration + 1, FLAGS, log)
        if hasattr(tr_model, "sync_batch_norm_state"):
            tr_model.sync_batch_norm_state()
        cv = eval_one_iteration(cv_model, cv_batches, valid_batch_per_iter, iteration + 1, FLAGS, log)
        end = datetime.datetime.now()
        log("{0}/{1} (INFO): d_learning_rate = {2:.5e}, g_learning_rate = {3:.5e}, time = {4:.3f} min\n"
            "{0}/{1} (TRAIN AVG.LOSS): {5}\n{0}/{1} (CROSS AVG.LOSS): {6}".format(
                iteration + 1, max_iters, tr_model.d_learning_rate, tr_model.g_learning_rate, (end - start).seconds / 60.0, _fmt(tr), _fmt(cv)))
        # Start decay learning rate (:462-474)
        tr_model.g_learning_rate = exponential_decay(iteration + 1, FLAGS.num_gpu, min_iters, FLAGS.g_learning_rate)
        tr_model.d_learning_rate = exponential_decay(iteration + 1, FLAGS.num_gpu, min_iters, FLAGS.d_learning_rate)
        tr_model.disc_noise_std = exponential_decay(iteration + 1, FLAGS.num_gpu, min_iters, FLAGS.init_disc_noise_std, multiply_jobs=False) \
            if FLAGS.init_disc_noise_std > 0 else 0.0        # (discriminator_dnn's noise layer is commented out, discriminator_dnn.py:58)
        cv_g_loss = cv[6]
        windows_g_loss.append(cv_g_loss)
        history.append(cv_g_loss)
        # Accept or reject new parameters (:479-498)
        if (iteration + 1) % CHECK_INTERVAL == 0:
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
        # Stopping criterion (:500-506)
        if iteration + 1 > min_iters and (iteration + 1) % CHECK_INTERVAL == 0:
            if g_rel_impr < FLAGS.end_improve:
                log("Iteration %d: Finished, too small relative G improvement %g" % (iteration + 1, g_rel_impr))
                break
    if windows_g_loss:                                         # the last, incomplete window (:509-521)
        g_loss_new = float(np.mean(windows_g_loss))
        g_rel_impr = (g_loss_prev - g_loss_new) / g_loss_prev
        if g_rel_impr > 0.0:
            tr_model.save(tr_model.save_dir, iteration + 1)
            log("Iteration {}: Nnet Accepted. Save model SUCCESS. g_loss_prev = {:.5f}, g_loss_new = {:.5f}".format(
                iteration + 1, g_loss_prev, g_loss_new))
    log("Training Done.")
    return history


def main(argv=None):
    FLAGS, unparsed = build_parser().parse_known_args(argv)
    rank, local, world = rdist.init_from_env()
    if world > 1:
        FLAGS.num_gpu = world
    if FLAGS.decode:
        # decode (:150-250) = the epoch script's, except that the checkpoint is read WITHOUT the moving averages (:205)
        rdist.run_on_rank0(lambda: decode(FLAGS, moving_average=False))
    else:
        train(FLAGS)


if __name__ == "__main__":
    main()
