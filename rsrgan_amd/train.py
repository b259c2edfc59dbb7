"""The caller of the hot path: scripts/train_gan_rnn_placeholder.py:48-201 and utils/ops.py:378-391."""
from __future__ import annotations

import math

import os
import numpy as np
import torch

from . import dist as rdist


def exponential_decay(iteration, num_jobs, num_ietrs, init_learning_rate, multiply_jobs=True):
    """utils/ops.py:378-391 (argument names as in the reference)."""
    final_learning_rate = 0.0001 * init_learning_rate
    if iteration + 1 >= num_ietrs:
        current_learning_rate = final_learning_rate
    else:
        current_learning_rate = init_learning_rate * math.exp(
            iteration * math.log(final_learning_rate / init_learning_rate) / num_ietrs)
    return num_jobs * current_learning_rate if multiply_jobs else current_learning_rate


def _batches(train_queue):
    if hasattr(train_queue, "get") and not isinstance(train_queue, (list, tuple)):
        while True:
            item = train_queue.get()
            if item is None:
                return
            yield item
    else:
        for item in train_queue:
            yield item


def _on_stream(model):
    import contextlib
    f = getattr(model, "on_stream", None)
    return f() if f is not None else contextlib.nullcontext()


def _check_device(model, d, g):
    """A persistent recurrence launch whose bounded wait expired poisons its losses with NaN (csrc/dpersist.hip): say so instead of
    handing non-finite numbers to the accept / reject logic of the outer loop."""
    if np.all(np.isfinite(d)) and np.all(np.isfinite(g)):
        return
    status = getattr(getattr(model, "engine", None), "device_status", None)
    code = status() if status is not None else 0
    if code:
        raise RuntimeError("persistent recurrence launch failed on the device: workgroup %d timed out waiting for a peer "
                           "(rsrgan_device_status; RSRGAN_DPERSIST=0 selects the per-step launches)" % (code - 1))


def train_one_iteration(sess, model, tr_num_batch, iteration, train_queue, num_gpu=None, share_g_forward=True):
    with _on_stream(model):          # the whole iteration on the engine's stream: no per-call stream hand-over
        return _train_one_iteration(sess, model, tr_num_batch, iteration, train_queue, num_gpu, share_g_forward)


def _train_one_iteration(sess, model, tr_num_batch, iteration, train_queue, num_gpu=None, share_g_forward=True):
    """Runs the model one iteration on given data (train_gan_rnn_placeholder.py:48-133).

    `train_queue` yields [ids, inputs, labels, lengths] (a Queue like the reference's, or any
    iterable).  Per batch: disc_updates D-runs then gen_updates G-runs on the same fed batch;
    batches whose row count is not batch_size*num_gpu are skipped (:69-70).  The batch is
    uploaded once (the reference feeds it 3x) and -- legal because G does not change between the
    last D update and the first G update -- the first G-run reuses the generator forward of the
    D-run unless share_g_forward=False.  Losses stay on the device until the end of the
    iteration.  Returns the same 7 averages."""
    num_gpu = num_gpu or model.num_gpu
    model.d_real = 1.0                                            # :63-64
    model.d_fake = 0.0
    d_counter = g_counter = 0
    d_acc = g_acc = None
    n_batches = int(tr_num_batch / num_gpu)
    it = _batches(train_queue)
    dev = model.engine.device
    for batch in range(n_batches):
        try:
            _, queue_inputs, queue_labels, queue_lengths = next(it)
        except StopIteration:
            break
        if queue_inputs.shape[0] != model.batch_size * num_gpu:   # :69-70
            continue
        x = model._shard(queue_inputs); lab = model._shard(queue_labels); ln = model._shard(queue_lengths)
        ready = getattr(model.engine, "upload_ready", None) if os.environ.get("RSRGAN_DPIPE", "0") not in ("", "0") else None
        later = getattr(model.engine, "upload_async", None)
        if ready is not None and not (isinstance(lab, torch.Tensor) and lab.device == dev):
            # RSRGAN_DPIPE=1: labels and lengths complete BEFORE the D-run is called (copied on the engine's upload stream while the
            # previous step still runs): D(real) of this batch then runs beside the previous G-run's tail
            lab = ready(lab); ln = ready(ln, int32=True)
        if later is not None:
            # everything else travels on the upload stream as well and is consumed in stream order (an event, no host wait): a copy
            # queued on the compute stream would sit between the previous G-run and this D-run
            x = later(x)
            if not (isinstance(lab, torch.Tensor) and lab.device == dev):
                lab = later(lab); ln = later(ln, int32=True)
        elif not isinstance(x, torch.Tensor):
            x = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
            if not isinstance(lab, torch.Tensor):
                lab = torch.from_numpy(np.ascontiguousarray(lab, np.float32)).to(dev)
                ln = torch.from_numpy(np.ascontiguousarray(ln)).to(dev).to(torch.int32)
        elif x.device != dev:                                     # page-locked staging tensors from io.prefetch: asynchronous DMA
            x = x.to(dev, non_blocking=True)
            if lab.device != dev:
                lab = lab.to(dev, non_blocking=True); ln = ln.to(dev, non_blocking=True).to(torch.int32)
        for d_step in range(model.disc_updates):
            tw = model.d_step(x, lab, ln, sync=False, gather=False)   # this tower's [1, 3]; towers are averaged once, below
            m = tw.mean(0)                                        # np.mean over towers (:85-87)
            d_acc = m if d_acc is None else d_acc + m
            d_counter += 1
        for g_step in range(model.gen_updates):
            reuse = share_g_forward and g_step == 0 and model.disc_updates > 0
            tw = model.g_step(x, lab, ln, reuse_g_forward=reuse, sync=False, gather=False)
            m = tw.mean(0)
            g_acc = m if g_acc is None else g_acc + m
            g_counter += 1
        if batch % 100 == 0 and getattr(model, "writer", None) is not None:        # Save summary (:116-122)
            model.writer.add_summary(model.run_summaries(x, lab, ln), iteration * tr_num_batch)
    # np.mean over towers (:85-87,104-107) commutes with the mean over steps: one all-reduce of the 7 sums per iteration
    if d_acc is not None:
        d_acc = rdist.all_reduce_mean_(d_acc.clone(), getattr(model, "process_group", None))
    if g_acc is not None:
        g_acc = rdist.all_reduce_mean_(g_acc.clone(), getattr(model, "process_group", None))
    d = (d_acc / max(d_counter, 1)).cpu().numpy() if d_acc is not None else np.zeros(3)
    g = (g_acc / max(g_counter, 1)).cpu().numpy() if g_acc is not None else np.zeros(4)
    _check_device(model, d, g)
    return float(d[0]), float(d[1]), float(d[2]), float(g[0]), float(g[1]), float(g[2]), float(g[3])


def eval_one_iteration(sess, model, cv_num_batch, iteration, valid_queue, num_gpu=None):
    with _on_stream(model):
        return _eval_one_iteration(sess, model, cv_num_batch, iteration, valid_queue, num_gpu)


def _eval_one_iteration(sess, model, cv_num_batch, iteration, valid_queue, num_gpu=None):
    """Cross validate the model on given data (train_gan_rnn_placeholder.py:136-201): the same
    fetches without the *_opt ops."""
    num_gpu = num_gpu or model.num_gpu
    model.d_real = 1.0
    model.d_fake = 0.0
    d_acc = g_acc = None
    n = 0
    it = _batches(valid_queue)
    for batch in range(int(cv_num_batch / num_gpu)):
        try:
            _, x, lab, ln = next(it)
        except StopIteration:
            break
        td = model.d_step(x, lab, ln, train=False, sync=False).mean(0)
        tg = model.g_step(x, lab, ln, train=False, sync=False).mean(0)
        d_acc = td if d_acc is None else d_acc + td
        g_acc = tg if g_acc is None else g_acc + tg
        n += 1
        last = (x, lab, ln)
    w = model.writer_for(False) if n and hasattr(model, "writer_for") else None
    if w is not None:                                                              # :186-190 (the last fed batch)
        w.add_summary(model.run_summaries(*last), iteration * cv_num_batch)
    d = (d_acc / max(n, 1)).cpu().numpy() if d_acc is not None else np.zeros(3)
    g = (g_acc / max(n, 1)).cpu().numpy() if g_acc is not None else np.zeros(4)
    _check_device(model, d, g)
    return float(d[0]), float(d[1]), float(d[2]), float(g[0]), float(g[1]), float(g[2]), float(g[3])
