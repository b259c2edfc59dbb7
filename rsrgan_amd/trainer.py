"""RNNTrainer / DNNTrainer -- host-side mirrors of the reference's generator-only trainers
(models/rnn_trainer.py:66-156, models/dnn_trainer.py:64-148): the same generator networks as the GAN models,
trained on g_loss = 0.5*output_dim*mse(G(x), labels) + l2 with Adam (+ per-tensor clip_by_norm 15 for the RNN
trainer only, rnn_trainer.py:137-138), EMA of all variables.  They run on the same librsrgan_hip.so entry points
with RSRGAN_FLAG_SUPERVISED: no discriminator pass is launched and rsrgan_d_step is an error.

The reference bakes tf.data tensors (`inputs, labels, lengths`) into the graph; here batches are passed to
`step()`, which returns the three per-tower lists the caller fetches (g_mse_losses, g_l2_losses, g_losses:
scripts/train_rnn.py / train_dnn.py fetch [model.g_opt, model.g_losses, ...])."""
from __future__ import annotations

from typing import Optional

from .gan import GAN
from .gan_rnn import GAN_RNN

FLAG_WAVEFRONT, FLAG_SUPERVISED = 1, 16


class _Args(object):
    def __init__(self, args, **kw):
        self.__dict__.update(vars(args) if not isinstance(args, dict) else args)
        self.__dict__.update(kw)


class RNNTrainer(GAN_RNN):
    """models/rnn_trainer.py:66 -- g_type in {lstm, res_lstm_l, res_lstm_base} (bnlstm / res_lstm_i are not built)."""

    def __init__(self, sess, args, devices, inputs=None, labels=None, lengths=None, cross_validation=False,
                 name="RNNTrainer", *, max_frames: Optional[int] = None, engine=None, process_group=None, seed: int = 4321,
                 net_overrides: Optional[dict] = None, share_engine_from=None):
        ov = dict(net_overrides or {})
        ov["flags"] = ov.get("flags", FLAG_WAVEFRONT) | FLAG_SUPERVISED
        super(RNNTrainer, self).__init__(sess, _Args(args, init_mse_weight=1.0), devices, cross_validation=cross_validation,
                                         name=name, max_frames=max_frames, engine=engine, process_group=process_group, seed=seed,
                                         net_overrides=ov, share_engine_from=share_engine_from)

    def d_step(self, *a, **k):
        raise RuntimeError("RNNTrainer has no discriminator (models/rnn_trainer.py)")

    def _summary_fetch(self, inputs, labels, lengths):
        """Model.run_summaries' five-value contract (d, g, x, labels, y) on THIS rank's rows: the engine directly, no collective
        (only the writer's rank gets here; g_step would all-gather the towers).  No discriminator: the three d_* scalars are 0."""
        x, lab, ln = self._shard(inputs), self._shard(labels), self._shard(lengths)
        with self.on_stream():
            g = self.engine.g_backward(x, lab, ln, None, train=False, reuse=False, apply=False)
            y = self.engine.forward_g(x, ln)
        return [0.0, 0.0, 0.0], g, x, lab, y

    def step(self, inputs, labels, lengths, train=True, sync=True):
        """sess.run([model.g_opt, model.g_mse_losses, model.g_l2_losses, model.g_losses])."""
        out = self.g_step(inputs, labels, lengths, train=train, sync=sync)
        return out if not sync else (out[1], out[2], out[3])


class DNNTrainer(GAN):
    """models/dnn_trainer.py:64 -- g_type 'dnn' or 'rced' (models/rced.py, batch_norm=False; 'cnn' is not built); no gradient
    clipping (dnn_trainer.py:124-126).  Conv kernels are exposed in TF's [S, fw, Cin, Cout] shape by get_vars/set_vars."""
    G_TYPES = ("dnn", "rced")

    def __init__(self, sess, args, devices, inputs=None, labels=None, cross_validation=False, name="DNNTrainer", *,
                 engine=None, process_group=None, seed: int = 4321, net_overrides: Optional[dict] = None):
        ov = dict(net_overrides or {})
        ov["flags"] = ov.get("flags", FLAG_WAVEFRONT) | FLAG_SUPERVISED
        super(DNNTrainer, self).__init__(sess, _Args(args, init_mse_weight=1.0), devices, cross_validation=cross_validation,
                                         name=name, engine=engine, process_group=process_group, seed=seed, net_overrides=ov)

    def d_step(self, *a, **k):
        raise RuntimeError("DNNTrainer has no discriminator (models/dnn_trainer.py)")

    def _summary_fetch(self, inputs, labels, lengths=None):
        """as RNNTrainer's: (d, g, x, labels, y) of the batch this rank drew, collective-free"""
        g = self.engine.g_backward(self._frames(inputs), self._frames(labels), None, train=False, reuse=False, apply=False)
        return [0.0, 0.0, 0.0], g, inputs, labels, self.forward(inputs)

    def step(self, inputs, labels, train=True, sync=True):
        out = self.g_step(inputs, labels, train=train, sync=sync)
        return out if not sync else (out[1], out[2], out[3])
