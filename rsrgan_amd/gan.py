"""GAN -- host-side mirror of the reference's frame-level GAN (models/gan.py:60-300): DNN generator
(models/dnn.py) + discriminator_dnn on concat(centre noisy frame, clean|enhanced MFCC), LSGAN losses with
constant targets 1/0, Adam for both nets, no gradient clipping (gan.py:125-143).  Frames are independent, so
the C ABI is driven with T = 1 and batch_size = frames per step."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import dist as rdist
from .gan_rnn import Model, NET_D, NET_G


class GAN(Model):
    G_TYPES = ("dnn",)            # models/gan.py:109-112; DNNTrainer widens this (dnn_trainer.py:94-101)

    """Generative Adversarial Network for Speech Enhancement (models/gan.py:60).  `inputs`/`labels` of the
    reference constructor are tf.data tensors; here batches are passed to d_step / g_step."""

    def __init__(self, sess, args, devices, inputs=None, labels=None, cross_validation=False, name="GAN", *,
                 engine=None, process_group=None, seed: int = 4321, net_overrides: Optional[dict] = None):
        super(GAN, self).__init__(name)
        self.sess, self.cross_validation = sess, cross_validation
        self.MOVING_AVERAGE_DECAY = 0.9999
        self.keep_prob = 1.0 if cross_validation else float(getattr(args, "keep_prob", 1.0))
        if not getattr(args, "l2_scale", 0.0) > 0.0:
            self.keep_prob = 1.0          # dnn.py:67-71, discriminator_dnn.py:47-51: reset unless l2_scale > 0 and is_training
        self.batch_norm = bool(getattr(args, "batch_norm", False))
        self.batch_size, self.devices = args.batch_size, devices
        self.num_gpu = getattr(args, "num_gpu", 1)
        self.save_dir = getattr(args, "save_dir", None)
        self.l2_scale = getattr(args, "l2_scale", 0.0)
        self.input_dim, self.output_dim = args.input_dim, args.output_dim
        self.left_context, self.right_context = getattr(args, "left_context", 0), getattr(args, "right_context", 0)
        self.g_disturb_weights = self.d_clip_weights = False
        self.disc_updates, self.gen_updates = getattr(args, "disc_updates", 1), getattr(args, "gen_updates", 1)
        if args.g_type not in self.G_TYPES:
            raise ValueError("Unrecognized G type {}".format(args.g_type))          # gan.py:111-112
        self.process_group = process_group
        fed = self.input_dim * (self.left_context + 1 + self.right_context)
        if engine is not None:
            self.engine = engine
        else:
            from .engine_hip import HipEngine
            self.engine = HipEngine(batch_size=self.batch_size, max_frames=1, input_dim=fed, output_dim=self.output_dim,
                                    g_type=args.g_type, d_type="dnn", d_joint_off=self.input_dim * self.left_context,
                                    g_splice=self.left_context + 1 + self.right_context,
                                    d_joint_dim=self.input_dim, l2_scale=self.l2_scale, cross_validation=cross_validation,
                                    batch_norm=self.batch_norm, seed=seed, **(net_overrides or {}))
        self._open_writer(args)
        if self.keep_prob < 1.0:              # every rank draws its own masks
            self.engine.set_dropout(self.keep_prob, seed + 0x9E3779B9 * rdist.rank(process_group))
        self.ema_enabled = getattr(self.engine, "ema_enabled", True)
        self._scalars = {}
        self.mse_lambda = getattr(args, "init_mse_weight", 10.0)
        self.d_learning_rate = getattr(args, "d_learning_rate", 1e-4)
        self.g_learning_rate = getattr(args, "g_learning_rate", 1e-4)

    def _set(self, k, v):
        self._scalars[k] = float(v)
        self.engine.set_scalar(k, float(v))

    d_learning_rate = property(lambda s: s._scalars["d_learning_rate"], lambda s, v: s._set("d_learning_rate", v))
    g_learning_rate = property(lambda s: s._scalars["g_learning_rate"], lambda s, v: s._set("g_learning_rate", v))
    mse_lambda = property(lambda s: s._scalars["mse_lambda"], lambda s, v: s._set("mse_lambda", v))

    @staticmethod
    def _frames(a):
        return a[:, None, :] if a.ndim == 2 else a

    def d_step(self, inputs, labels, train=True, sync=True):
        """sess.run([model.d_opt, model.d_rl_losses, model.d_fk_losses, model.d_losses])."""
        train = train and not self.cross_validation
        x, lab = self._frames(inputs), self._frames(labels)
        if train and rdist.world_size(self.process_group) > 1:
            losses = self.engine.d_backward(x, lab, None, train=True, apply=False)
            self._average_gradients(NET_D)
            self.engine.apply(NET_D)
        else:
            losses = self.engine.d_backward(x, lab, None, train=train, apply=train)
        tw = rdist.all_gather_rows(losses, self.process_group)
        if not sync:
            return tw
        tw = tw.cpu().numpy()
        return list(tw[:, 0]), list(tw[:, 1]), list(tw[:, 2])

    def g_step(self, inputs, labels, train=True, reuse_g_forward=False, sync=True):
        """sess.run([model.g_opt, model.g_adv_losses, model.g_mse_losses, model.g_l2_losses, model.g_losses])."""
        train = train and not self.cross_validation
        x, lab = self._frames(inputs), self._frames(labels)
        if train and rdist.world_size(self.process_group) > 1:
            losses = self.engine.g_backward(x, lab, None, train=True, reuse=reuse_g_forward, apply=False)
            self._average_gradients(NET_G)
            self.engine.apply(NET_G)
        else:
            losses = self.engine.g_backward(x, lab, None, train=train, reuse=reuse_g_forward, apply=train)
        tw = rdist.all_gather_rows(losses, self.process_group)
        if not sync:
            return tw
        tw = tw.cpu().numpy()
        return list(tw[:, 0]), list(tw[:, 1]), list(tw[:, 2]), list(tw[:, 3])

    BN_STATE = ("moving_mean", "moving_variance", "renorm_mean", "renorm_mean_weight", "renorm_stddev", "renorm_stddev_weight")

    def sync_batch_norm_state(self):
        """The reference's towers update ONE copy of the batch-norm statistics (models/gan.py:139-146: the update ops of every
        tower act on shared variables); here every rank commits the statistics of its own batches, so they drift apart and a
        checkpoint holds rank 0's.  This puts the mean over ranks into every rank's copy (the outer loops call it once per epoch /
        iteration, before the cross-validation pass and the checkpoint).  Returns the number of floats exchanged."""
        if not self.batch_norm or rdist.world_size(self.process_group) <= 1:
            return 0
        n = 0
        for net in (NET_G, NET_D):
            ranges = [(off, int(np.prod(shape))) for name, shape, off in self.engine.tensor_table(net)
                      if "/BatchNorm/" in name and name.rsplit("/", 1)[1] in self.BN_STATE]
            if not ranges:
                continue
            flat = self.engine.get_params(net, "variables")
            packed = torch.cat([flat[o:o + c] for o, c in ranges])
            rdist.all_reduce_mean_(packed, self.process_group)
            at = 0
            for o, c in ranges:
                flat[o:o + c] = packed[at:at + c]
                at += c
            self.engine.set_params(net, flat, "variables")
            n += at
        return n

    def _summary_fetch(self, inputs, labels, lengths=None):
        """collective-free (only the writer's rank calls it): the engine directly, on the batch this rank drew"""
        x, lab = self._frames(inputs), self._frames(labels)
        d = self.engine.d_backward(x, lab, None, train=False, apply=False)
        g = self.engine.g_backward(x, lab, None, train=False, reuse=False, apply=False)
        return d, g, inputs, labels, self.forward(inputs)

    def forward(self, inputs):
        """sess.run(model.generator outputs): enhanced MFCC frames [N, output_dim]."""
        y = self.engine.forward_g(self._frames(inputs), None)
        y = y[:, 0, :]
        return y.cpu().numpy() if not isinstance(inputs, torch.Tensor) else y

    _RCED_WIDTHS = (13, 11, 9, 7, 7, 7, 9, 11, 13)                     # models/rced.py:93

    def _tf_shape(self, name, shape):
        """The library keeps a conv2d kernel as the [S*fw*Cin, Cout] GEMM operand; TF's variable is [S, fw, Cin, Cout]."""
        if "/Conv" in name and name.endswith("/weights"):
            idx = name.split("/Conv")[1].split("/")[0]
            fw = self._RCED_WIDTHS[int(idx[1:]) if idx else 0]
            S = self.left_context + 1 + self.right_context
            return (S, fw, shape[0] // (S * fw), shape[1])
        if name.endswith("_weight") and "/BatchNorm/renorm_" in name:      # renorm_mean_weight / renorm_stddev_weight: shape ()
            return ()
        return tuple(shape)

    def get_vars(self):
        out = []
        for net, pre in ((NET_G, "g_"), (NET_D, "d_")):
            flat = self.engine.get_params(net, "variables").cpu().numpy()
            d = {}
            for name, shape, off in self.engine.tensor_table(net):
                assert name.startswith(pre), name
                d[name] = flat[off:off + int(np.prod(shape))].reshape(self._tf_shape(name, shape))
            out.append(d)
        return out[0], out[1]

    def set_vars(self, g_vars=None, d_vars=None, reset_ema=True):
        for net, vals in ((NET_G, g_vars), (NET_D, d_vars)):
            if vals is None:
                continue
            flat = np.zeros(self.engine.param_count(net), np.float32)
            for name, shape, off in self.engine.tensor_table(net):
                v = np.asarray(vals[name], np.float32)
                assert tuple(v.shape) == self._tf_shape(name, shape), (name, v.shape, shape)
                flat[off:off + v.size] = v.reshape(-1)
            self.engine.set_params(net, flat, "variables")
            if reset_ema and self.ema_enabled:
                self.engine.set_params(net, flat, "ema")
