"""Feature pipeline between the ark reader and GAN_RNN.d_step/g_step: global CMVN
(io_funcs/make_tfrecords.py:84-87), Kaldi-style splicing (io_funcs/tfrecords_io.py:177-203) and
length-bucketed, zero-padded batches (io_funcs/tfrecords_dataset.py:139-175)."""
from __future__ import annotations

import random
from typing import Iterator, List, Optional

import numpy as np

from .kaldi_ark import ArkReader


def apply_cmvn(inputs, labels, cmvn):
    """make_tfrecords.py:84-87: float64 arithmetic, (x - mean) / stddev."""
    inputs = np.array(inputs, np.float64)                 # private float64 copy, then in place (no temporaries)
    inputs -= cmvn["mean_inputs"]; inputs /= cmvn["stddev_inputs"]
    if labels is not None:
        labels = np.array(labels, np.float64)
        labels -= cmvn["mean_labels"]; labels /= cmvn["stddev_labels"]
    return inputs, labels


def splice_feats(feats, left, right):
    """tfrecords_io.py:177-203: [row, col] -> [row, col*(left+1+right)].  The i-th left context is
    feats shifted down by i with the first row repeated (tf.pad SYMMETRIC one row at a time), the
    i-th right context is feats shifted up by i with the last row repeated; order: left far..near,
    centre, right near..far."""
    feats = np.asarray(feats)
    row = feats.shape[0]
    idx = np.arange(row)
    parts = [feats[np.maximum(idx - i, 0)] for i in range(left, 0, -1)]
    parts.append(feats)
    parts += [feats[np.minimum(idx + i, row - 1)] for i in range(1, right + 1)]
    return np.concatenate(parts, 1)


class PaddedBatchReader(object):
    """get_padded_batch (tfrecords_dataset.py:53-180) without TFRecords: reads (inputs, labels)
    utterance pairs from two scp files, applies CMVN and splicing, groups utterances into windows of
    `batch_size` by length bucket (start 200 frames, width 50, ids clipped at num_buckets, :157-171),
    pads each batch with zeros to its longest utterance and yields
    [utt_ids, inputs [B,T,D*(L+1+R)] f32, labels [B,T,Dout] f32, lengths [B] i32] -- the items
    train_one_iteration pops from its queue (train_gan_rnn_placeholder.py:67).  Partial windows are
    emitted at the end of the data, like tf.contrib.data.group_by_window; the training loop skips them."""

    def __init__(self, inputs_scp, labels_scp, batch_size, left_context=0, right_context=0, cmvn=None,
                 num_buckets=20, shuffle=True, seed=None):
        self.inputs, self.labels = ArkReader(), ArkReader()
        self.inputs(inputs_scp)
        self.labels(labels_scp)
        assert self.inputs.utt_ids == self.labels.utt_ids, "inputs_utt_id == labels_utt_id (make_tfrecords.py:35)"
        self.batch_size, self.left, self.right = batch_size, left_context, right_context
        self.cmvn, self.num_buckets, self.shuffle = cmvn, num_buckets, shuffle
        self.rng = random.Random(seed)
        self._nframes = None

    def __len__(self):
        return len(self.inputs.utt_ids)

    def _utt(self, i):
        x = self.inputs.read_utt_data_from_index(i).astype(np.float64)
        y = self.labels.read_utt_data_from_index(i).astype(np.float64)
        if self.cmvn is not None:
            x, y = apply_cmvn(x, y, self.cmvn)
        return self.inputs.utt_ids[i], splice_feats(x, self.left, self.right).astype(np.float32), y.astype(np.float32)

    def _pad(self, items):
        T = max(x.shape[0] for _, x, _ in items)
        ids = [u for u, _, _ in items]
        X = np.empty((len(items), T, items[0][1].shape[1]), np.float32)       # every element is written exactly once below
        Y = np.empty((len(items), T, items[0][2].shape[1]), np.float32)
        L = np.zeros(len(items), np.int32)
        for b, (_, x, y) in enumerate(items):
            n = x.shape[0]
            X[b, :n] = x; X[b, n:] = 0.0
            Y[b, :n] = y; Y[b, n:] = 0.0
            L[b] = n
        return [ids, X, Y, L]

    def plan(self) -> Iterator[List[int]]:
        """The utterance indices of every batch, in the order __iter__ yields them -- from the matrix headers alone, so the
        expensive part (materialize) can run on several threads (io.prefetch)."""
        if self._nframes is None:
            self._nframes = [self.inputs.utt_shape_from_index(i)[0] for i in range(len(self))]
        order = list(range(len(self)))
        if self.shuffle:
            self.rng.shuffle(order)
        windows = {}
        for i in order:
            if self.num_buckets > 1:
                key = min(self.num_buckets, (self._nframes[i] - 200) // 50)      # :157-165
            else:
                key = 0
            w = windows.setdefault(key, [])
            w.append(i)
            if len(w) == self.batch_size:
                yield w
                windows[key] = []
        for key in sorted(windows):
            if windows[key]:
                yield windows[key]

    def materialize(self, indices: List[int]) -> List:
        return self._pad([self._utt(i) for i in indices])

    def __iter__(self) -> Iterator[List]:
        for indices in self.plan():
            yield self.materialize(indices)


class FrameBatchReader(object):
    """get_batch of the frame-level recipes (io_funcs/tfrecords_io.py:206-255) without TFRecords: every utterance is normalised,
    spliced and its frames `enqueue_many`-ed into a tf.RandomShuffleQueue(capacity = 1000 + (num_threads + 1) * batch_size,
    min_after_dequeue = 1000); `dequeue_many(batch_size)` draws that many frames uniformly at random from the queue.  Yields
    [inputs [N, D*(L+1+R)] f32, labels [N, Dout] f32].  Utterances are visited in shuffled order once per pass
    (string_input_producer(shuffle=True, num_epochs)); the frames left over at the end of the data that do not fill a batch are
    dropped, as dequeue_many does when the queue closes."""

    def __init__(self, inputs_scp, labels_scp, batch_size, left_context=0, right_context=0, cmvn=None, num_threads=4,
                 shuffle=True, seed=None):
        self.inputs, self.labels = ArkReader(), ArkReader()
        self.inputs(inputs_scp)
        self.labels(labels_scp)
        assert self.inputs.utt_ids == self.labels.utt_ids, "inputs_utt_id == labels_utt_id (make_tfrecords.py:35)"
        self.batch_size, self.left, self.right, self.cmvn, self.shuffle = batch_size, left_context, right_context, cmvn, shuffle
        self.capacity = 1000 + (num_threads + 1) * batch_size
        self.min_after_dequeue = 1000 if shuffle else 0
        self.rng = np.random.default_rng(seed)

    def num_frames(self):
        return sum(self.inputs.utt_shape_from_index(i)[0] for i in range(len(self.inputs.utt_ids)))

    def num_batches(self):
        """get_num_batch (train_gan_dnn.py:346-370) counts dequeued batches of one pass: floor(frames / batch_size)."""
        return self.num_frames() // self.batch_size

    def _utt(self, i):
        x = self.inputs.read_utt_data_from_index(i).astype(np.float64)
        y = self.labels.read_utt_data_from_index(i).astype(np.float64)
        if self.cmvn is not None:
            x, y = apply_cmvn(x, y, self.cmvn)
        return splice_feats(x, self.left, self.right).astype(np.float32), y.astype(np.float32)

    def __iter__(self):
        order = np.arange(len(self.inputs.utt_ids))
        if self.shuffle:
            self.rng.shuffle(order)
        qx, qy, held = [], [], 0                   # the queue as a list of per-utterance blocks + how many frames it holds
        todo = list(order)
        N = self.batch_size

        def draw():
            nonlocal qx, qy, held
            X, Y = np.concatenate(qx), np.concatenate(qy)
            pick = self.rng.choice(held, N, replace=False) if self.shuffle else np.arange(N)
            keep = np.ones(held, bool); keep[pick] = False
            bx, by = X[pick], Y[pick]
            qx, qy, held = [X[keep]], [Y[keep]], held - N
            return [bx, by]
        while todo or held >= N:
            # the enqueuing threads keep the queue as full as its capacity allows: an utterance is enqueued when it fits, and
            # whenever no batch can be drawn (held < N) -- tf's enqueue_many admits an over-long utterance piecewise as room
            # frees up, so it never blocks the dequeue side for good (an utterance longer than the capacity must not hang)
            while todo and (held < N or held + self.inputs.utt_shape_from_index(int(todo[0]))[0] <= self.capacity):
                x, y = self._utt(int(todo.pop(0)))
                qx.append(x); qy.append(y); held += x.shape[0]
            if held >= N and (held - N >= self.min_after_dequeue or not todo):
                yield draw()
            elif not todo:
                break
            elif held >= N:                        # an utterance longer than the room left: dequeue first, as a blocked enqueue would wait
                yield draw()
