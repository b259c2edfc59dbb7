"""GAN_RNN -- host-side mirror of the reference's model object for the sequence-level GAN step.

Same constructor arguments, attributes, scalars and save/load as
models/gan_rnn_placeholder.py:GAN_RNN (:62-298), but one *process* per GPU instead of in-graph
towers: the per-tower gradient mean (utils/ops.py:343-376 average_gradients) becomes an RCCL
all-reduce(avg) over xGMI, after which the per-tensor clip and the SGD/Adam update run exactly
as in :177-184.  `sess.run([model.d_opt, ...], feed)` becomes `model.d_step(inputs, labels,
lengths)`, `sess.run([model.g_opt, ...], feed)` becomes `model.g_step(...)`.

The compute engine defaults to the HIP library; there is no CPU fallback.
"""
from __future__ import annotations

import contextlib
import glob
import os
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import dist as rdist

NET_G, NET_D = 0, 1


def _bn_statistic(name):
    return "/BatchNorm/" in name and name.rsplit("/", 1)[-1] not in ("beta", "gamma")


class Model(object):
    """Base class (gan_rnn_placeholder.py:21-60): save / load with tf.train.Saver semantics
    (max_to_keep=10, `checkpoint` state file naming the latest), stored as .npz keyed by the
    reference's variable names."""

    def __init__(self, name="BaseModel"):
        self.name = name

    def _average_gradients(self, net):
        """average_gradients over towers (utils/ops.py:343-376) = all-reduce(mean) over ranks; the HIP engine does it bucket
        by bucket overlapped with the rest of the backward, any other engine (tests) as one all-reduce of the buffer."""
        if hasattr(self.engine, "all_reduce_grads"):
            self.engine.all_reduce_grads(net, self.process_group)
        else:
            rdist.all_reduce_mean_(self.engine.grad_view(net), self.process_group)

    # -- TensorBoard summaries (gan_rnn_placeholder.py:81-86,219-223,270-298; gan.py:77-82,184-250) -------------------
    writer = None
    _eval_writer = None

    def _open_writer(self, args):
        """tf.summary.FileWriter(save_dir/train) -- or /eval on a cross_validation model; rank 0 only; args.write_summaries=False
        turns it off"""
        if self.save_dir and getattr(args, "write_summaries", True) and rdist.rank(getattr(self, "process_group", None)) == 0:
            from .summary import FileWriter
            self.writer = FileWriter(os.path.join(self.save_dir, "eval" if self.cross_validation else "train"))

    def writer_for(self, train):
        """the outer loops here fetch the cross-validation twin's values from the training model (train=False): its events go to
        save_dir/eval like the twin's"""
        if train or self.writer is None or self.cross_validation:
            return self.writer
        if self._eval_writer is None:
            from .summary import FileWriter
            self._eval_writer = FileWriter(os.path.join(self.save_dir, "eval"))
        return self._eval_writer

    def run_summaries(self, inputs, labels, lengths=None):
        """sess.run(model.summaries, feed) (train_gan_rnn_placeholder.py:118-121, train_gan_dnn.py:133): the seven loss scalars of
        the fed batch (fetched without the *_opt ops) + histograms of the batch and of the generator's output, serialised as a
        Summary for writer.add_summary."""
        from .summary import model_summaries
        # Only the rank that owns the writer (rank 0) gets here, so the fetch must not contain a collective: _summary_fetch works
        # on this rank's shard of the fed batch and calls the engine directly (d_step / g_step gather the towers' losses, and the
        # other ranks are already in their next gradient all-reduce).  The summary describes tower 0, as the reference's
        # tf.summary ops do (gan_rnn_placeholder.py:219-223 are built in tower 0's name scope).
        d, g, x, lab, y = self._summary_fetch(inputs, labels, lengths)
        host = lambda a: a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
        return model_summaries([float(v) for v in list(host(d).reshape(-1)) + list(host(g).reshape(-1))], host(x), host(lab), host(y))

    def save(self, save_dir, step):
        """rank 0 writes; every rank returns once the checkpoint exists (or raises if rank 0 could not write it)"""
        os.makedirs(save_dir, exist_ok=True)
        return rdist.run_on_rank0(lambda: self._save_rank0(save_dir, step), self.process_group)

    def _save_rank0(self, save_dir, step):
        base = "%s-%d" % (self.name, int(step))
        payload = {}
        for net, tag in ((NET_G, ""), (NET_D, "")):
            flat = self.engine.get_params(net, "variables").cpu().numpy()
            ema = self.engine.get_params(net, "ema").cpu().numpy() if self.ema_enabled else None
            for name, shape, off in self.engine.tensor_table(net):
                n = int(np.prod(shape))
                payload[name] = flat[off:off + n].reshape(shape)
                if ema is not None:
                    payload[name + "/ExponentialMovingAverage"] = ema[off:off + n].reshape(shape)
        for what in ("adam_m", "adam_v"):
            flat = self.engine.get_params(NET_G, what).cpu().numpy()
            for name, shape, off in self.engine.tensor_table(NET_G):
                n = int(np.prod(shape))
                payload[name + ("/Adam" if what == "adam_m" else "/Adam_1")] = flat[off:off + n].reshape(shape)
        payload["__adam_step__"] = np.asarray(self.engine.get_scalar("adam_step"))
        if getattr(self.engine, "d_has_adam", False):              # models/gan.py: Adam for D as well
            for what in ("adam_m", "adam_v"):
                flat = self.engine.get_params(NET_D, what).cpu().numpy()
                for name, shape, off in self.engine.tensor_table(NET_D):
                    payload[name + ("/Adam" if what == "adam_m" else "/Adam_1")] = flat[off:off + int(np.prod(shape))].reshape(shape)
            payload["__adam_step_d__"] = np.asarray(self.engine.get_scalar("adam_step_d"))
        path = os.path.join(save_dir, base + ".npz")
        np.savez(path, **payload)
        kept = sorted(glob.glob(os.path.join(save_dir, self.name + "-*.npz")), key=os.path.getmtime)
        for old in kept[:-10]:                                     # Saver(max_to_keep=10)
            os.remove(old)
        with open(os.path.join(save_dir, "checkpoint"), "w") as f:
            f.write('model_checkpoint_path: "%s"\n' % base)
        return path

    def load(self, save_dir, model_file=None, moving_average=False):
        if not os.path.exists(save_dir):
            print("[!] Checkpoints path does not exist...")
            return False
        print("[*] Reading checkpoints...")
        if model_file is None:
            state = os.path.join(save_dir, "checkpoint")
            if not os.path.exists(state):
                return False
            with open(state) as f:
                line = f.readline()
            ckpt_name = line.split('"')[1] if '"' in line else ""
            if not ckpt_name:
                return False
        else:
            ckpt_name = model_file
        path = os.path.join(save_dir, ckpt_name if ckpt_name.endswith(".npz") else ckpt_name + ".npz")
        if not os.path.exists(path):
            return False
        data = np.load(path)
        if moving_average and not all((name + "/ExponentialMovingAverage") in data
                                      for net in (NET_G, NET_D) for name, _, _ in self.engine.tensor_table(net)):
            print("[!] {} holds no ExponentialMovingAverage variables".format(ckpt_name))
            return False
        for net in (NET_G, NET_D):
            table = self.engine.tensor_table(net)
            flat = np.zeros(self.engine.param_count(net), np.float32)
            for name, shape, off in table:
                # (batch-norm statistics are no trainable variables: variable_averages.apply(g_vars) has no shadow of them)
                key = name + "/ExponentialMovingAverage" if moving_average and not _bn_statistic(name) else name
                flat[off:off + int(np.prod(shape))] = data[key].reshape(-1)
            self.engine.set_params(net, flat, "variables")
            if not moving_average and self.ema_enabled and (table[0][0] + "/ExponentialMovingAverage") in data:
                for name, shape, off in table:
                    flat[off:off + int(np.prod(shape))] = data[name + "/ExponentialMovingAverage"].reshape(-1)
                self.engine.set_params(net, flat, "ema")
        if not moving_average:
            table = self.engine.tensor_table(NET_G)
            for what, suffix in (("adam_m", "/Adam"), ("adam_v", "/Adam_1")):
                flat = np.zeros(self.engine.param_count(NET_G), np.float32)
                for name, shape, off in table:
                    flat[off:off + int(np.prod(shape))] = data[name + suffix].reshape(-1)
                self.engine.set_params(NET_G, flat, what)
            self.engine.set_scalar("adam_step", float(data["__adam_step__"]))
            if getattr(self.engine, "d_has_adam", False) and "__adam_step_d__" in data:
                table = self.engine.tensor_table(NET_D)
                for what, suffix in (("adam_m", "/Adam"), ("adam_v", "/Adam_1")):
                    flat = np.zeros(self.engine.param_count(NET_D), np.float32)
                    for name, shape, off in table:
                        flat[off:off + int(np.prod(shape))] = data[name + suffix].reshape(-1)
                    self.engine.set_params(NET_D, flat, what)
                self.engine.set_scalar("adam_step_d", float(data["__adam_step_d__"]))
        print("[*] Read {}".format(ckpt_name))
        return True


class GAN_RNN(Model):
    """Generative Adversarial Network for Speech Enhancement (gan_rnn_placeholder.py:62).

    `sess` is accepted and ignored (there is no session); `devices` is the list the reference
    iterates over (:153) -- here exactly one entry per process.  Extra keyword-only arguments
    are the knobs the reference hard-codes or has no notion of."""

    def __init__(self, sess, args, devices, cross_validation=False, infer=False, name="GAN_RNN", *,
                 max_frames: Optional[int] = None, engine=None, process_group=None, seed: int = 4321,
                 net_overrides: Optional[dict] = None, share_engine_from: Optional["GAN_RNN"] = None):
        super(GAN_RNN, self).__init__(name)
        self.sess = sess
        self.cross_validation = cross_validation
        self.MOVING_AVERAGE_DECAY = 0.9999
        self.max_grad_norm = 15
        self.keep_prob = 1.0 if cross_validation else getattr(args, "keep_prob", 1.0)
        self.batch_norm = getattr(args, "batch_norm", False)
        if self.batch_norm:
            raise NotImplementedError("batch_norm is off on this path "
                                      "(run_gan_rnn_placeholder.sh:131; train_gan_rnn_placeholder.py:723-728)")
        self.batch_size = args.batch_size
        self.devices = devices
        self.num_gpu = getattr(args, "num_gpu", 1)
        self.save_dir = getattr(args, "save_dir", None)
        self.l2_scale = getattr(args, "l2_scale", 0.0)
        self.input_dim = args.input_dim
        self.output_dim = args.output_dim
        self.left_context = getattr(args, "left_context", 0)
        self.right_context = getattr(args, "right_context", 0)
        self.g_disturb_weights = False
        self.disc_updates = getattr(args, "disc_updates", 1)
        self.gen_updates = getattr(args, "gen_updates", 1)
        self.d_clip_weights = False
        self.g_type = args.g_type
        if self.g_type not in ("lstm", "res_lstm_l", "res_lstm_base"):
            raise ValueError("Unrecognized G type {}".format(self.g_type))      # :131-132
        self.process_group = process_group
        self.infer = infer
        din = self.input_dim * (self.left_context + 1 + self.right_context)
        if share_engine_from is not None:                    # the cross_validation twin shares variables (:436-437)
            self.engine = share_engine_from.engine
        elif engine is not None:
            self.engine = engine
        else:
            from .engine_hip import HipEngine                # raises if the HIP library / GPU is missing
            self.engine = HipEngine(batch_size=self.batch_size, max_frames=max_frames or 1000, input_dim=din,
                                    output_dim=self.output_dim, g_type=self.g_type, l2_scale=self.l2_scale,
                                    cross_validation=cross_validation, seed=seed, **(net_overrides or {}))
        self.ema_enabled = getattr(self.engine, "ema_enabled", True)
        self._scalars = {}
        if share_engine_from is None:
            self.mse_lambda = getattr(args, "init_mse_weight", 10.0)
            self.d_learning_rate = getattr(args, "d_learning_rate", 1e-3)
            self.g_learning_rate = getattr(args, "g_learning_rate", 8e-5)
            self.d_real = 1.0
            self.d_fake = 0.0
        else:
            self._scalars = share_engine_from._scalars
        self.disc_noise_std = getattr(args, "init_disc_noise_std", 0.0)
        if self.keep_prob < 1.0 and share_engine_from is None:
            # DropoutWrapper(output_keep_prob) around the generator's cells (lstm.py:99-102, res_lstm_l.py:96-99, res_lstm_base.py:96-99); every rank its own masks
            self.engine.set_dropout(self.keep_prob, seed + 0x9E3779B9 * rdist.rank(process_group))
        if share_engine_from is None or cross_validation:
            self._open_writer(args)
        self._noise_gen = None

    # -- mutable scalars: sess.run(tf.assign(model.<x>, v)) --------------------------------
    def _set(self, k, v):
        self._scalars[k] = float(v)
        self.engine.set_scalar(k, float(v))

    d_learning_rate = property(lambda s: s._scalars["d_learning_rate"], lambda s, v: s._set("d_learning_rate", v))
    g_learning_rate = property(lambda s: s._scalars["g_learning_rate"], lambda s, v: s._set("g_learning_rate", v))
    mse_lambda = property(lambda s: s._scalars["mse_lambda"], lambda s, v: s._set("mse_lambda", v))
    d_real = property(lambda s: s._scalars["d_real"], lambda s, v: s._set("d_real", v))
    d_fake = property(lambda s: s._scalars["d_fake"], lambda s, v: s._set("d_fake", v))

    def assign(self, name, value):
        setattr(self, name, value)

    # -- helpers -----------------------------------------------------------------------
    def _draw_noise(self):
        """gaussian_noise_layer (utils/ops.py:19-30): N(0, std^2) of shape [B,1,Dout]."""
        if self.disc_noise_std <= 0.0:
            return None
        dev = self.engine.device
        if self._noise_gen is None:
            self._noise_gen = torch.Generator(device=dev)
            self._noise_gen.manual_seed(1234 + rdist.rank(self.process_group))
        return torch.randn(self.batch_size, 1, self.output_dim, generator=self._noise_gen, device=dev) * self.disc_noise_std

    def _shard(self, a):
        """per-GPU batch slicing (:157-159): a fed [B*num_gpu, ...] batch -> this rank's rows."""
        if a is None:
            return None
        n = a.shape[0]
        if n == self.batch_size:
            return a
        ws, r = rdist.world_size(self.process_group), rdist.rank(self.process_group)
        if n != self.batch_size * ws:
            raise ValueError("batch has %d rows, expected %d or %d" % (n, self.batch_size, self.batch_size * ws))
        return a[self.batch_size * r:self.batch_size * (r + 1)]

    def on_stream(self):
        """The engine's own HIP stream as the current torch stream (hipGraph replay needs a real stream; a loop that stays
        under this context pays no null-stream hand-over per call).  A CPU test engine has no stream: no-op."""
        f = getattr(self.engine, "on_stream", None)
        return f() if f is not None else contextlib.nullcontext()

    def _towers(self, losses: torch.Tensor, gather: bool = True) -> torch.Tensor:
        """[k] device tensor -> [world, k]: the per-tower loss lists of :262-268.  gather=False keeps this rank's row only
        ([1, k], no collective): train_one_iteration averages over towers once per iteration instead of once per step."""
        if not gather:
            return losses.unsqueeze(0)
        return rdist.all_gather_rows(losses, self.process_group)

    # -- the two sess.run calls ------------------------------------------------------------
    def d_step(self, inputs, labels, lengths, noise_real=None, noise_fake=None, train=True, sync=True, gather=True):
        """sess.run([model.d_opt, model.d_rl_losses, model.d_fk_losses, model.d_losses], feed)
        (train_gan_rnn_placeholder.py:77-82).  Returns three per-tower lists (or, with
        sync=False, a [towers,3] device tensor)."""
        with self.on_stream():
            return self._d_step(inputs, labels, lengths, noise_real, noise_fake, train, sync, gather)

    def _d_step(self, inputs, labels, lengths, noise_real, noise_fake, train, sync, gather):
        x, lab, ln = self._shard(inputs), self._shard(labels), self._shard(lengths)
        nr = self._shard(noise_real) if noise_real is not None else self._draw_noise()
        nf = self._shard(noise_fake) if noise_fake is not None else self._draw_noise()
        train = train and not self.cross_validation
        ws = rdist.world_size(self.process_group)
        if train and ws > 1:
            losses = self.engine.d_backward(x, lab, ln, nr, nf, train=True, apply=False)
            self._average_gradients(NET_D)     # average_gradients
            self.engine.apply(NET_D)
        else:
            losses = self.engine.d_backward(x, lab, ln, nr, nf, train=train, apply=train)
        tw = self._towers(losses, gather or sync)
        if not sync:
            return tw
        tw = tw.cpu().numpy()
        return list(tw[:, 0]), list(tw[:, 1]), list(tw[:, 2])

    def g_step(self, inputs, labels, lengths, noise_fake=None, train=True, reuse_g_forward=False, sync=True, gather=True):
        """sess.run([model.g_opt, model.g_adv_losses, model.g_mse_losses, model.g_l2_losses,
        model.g_losses], feed) (train_gan_rnn_placeholder.py:94-101)."""
        with self.on_stream():
            return self._g_step(inputs, labels, lengths, noise_fake, train, reuse_g_forward, sync, gather)

    def _g_step(self, inputs, labels, lengths, noise_fake, train, reuse_g_forward, sync, gather):
        x, lab, ln = self._shard(inputs), self._shard(labels), self._shard(lengths)
        nf = self._shard(noise_fake) if noise_fake is not None else self._draw_noise()
        train = train and not self.cross_validation
        ws = rdist.world_size(self.process_group)
        if train and ws > 1:
            losses = self.engine.g_backward(x, lab, ln, nf, train=True, reuse=reuse_g_forward, apply=False)
            self._average_gradients(NET_G)
            self.engine.apply(NET_G)
        else:
            losses = self.engine.g_backward(x, lab, ln, nf, train=train, reuse=reuse_g_forward, apply=train)
        tw = self._towers(losses, gather or sync)
        if not sync:
            return tw
        tw = tw.cpu().numpy()
        return list(tw[:, 0]), list(tw[:, 1]), list(tw[:, 2]), list(tw[:, 3])

    def _summary_fetch(self, inputs, labels, lengths):
        """the 3 + 4 losses and G(x) of THIS rank's rows, without the *_opt ops and without any collective"""
        x, lab, ln = self._shard(inputs), self._shard(labels), self._shard(lengths)
        with self.on_stream():
            d = self.engine.d_backward(x, lab, ln, self._draw_noise(), self._draw_noise(), train=False, apply=False)
            g = self.engine.g_backward(x, lab, ln, self._draw_noise(), train=False, reuse=False, apply=False)
            y = self.engine.forward_g(x, ln)
        return d, g, x, lab, y

    def forward(self, inputs, lengths):
        """sess.run(model.g_outputs, {inputs, lengths}) (train_gan_rnn_placeholder.py:282-285)."""
        y = self.engine.forward_g(inputs, lengths)
        return y.cpu().numpy() if not isinstance(inputs, torch.Tensor) else y

    # -- variables ---------------------------------------------------------------------
    def get_vars(self):
        """d_vars / g_vars split by name prefix with the reference's asserts (:301-317)."""
        self.g_vars_dict, self.d_vars_dict = {}, {}
        for net, dst, pre in ((NET_G, self.g_vars_dict, "g_"), (NET_D, self.d_vars_dict, "d_")):
            flat = self.engine.get_params(net, "variables").cpu().numpy()
            for name, shape, off in self.engine.tensor_table(net):
                assert name.startswith(pre), name
                dst[name] = flat[off:off + int(np.prod(shape))].reshape(shape)
        return self.g_vars_dict, self.d_vars_dict

    def set_vars(self, g_vars=None, d_vars=None, reset_ema=True):
        """Inject variable values (parity tests / external initialisers).  reset_ema: the EMA shadow
        restarts from the injected values, as ExponentialMovingAverage.apply initialises it from the
        variable's initial value (:185-186)."""
        for net, vals in ((NET_G, g_vars), (NET_D, d_vars)):
            if vals is None:
                continue
            flat = np.zeros(self.engine.param_count(net), np.float32)
            for name, shape, off in self.engine.tensor_table(net):
                v = np.asarray(vals[name], np.float32)
                assert tuple(v.shape) == tuple(shape), (name, v.shape, shape)
                flat[off:off + v.size] = v.reshape(-1)
            self.engine.set_params(net, flat, "variables")
            if reset_ema and self.ema_enabled:
                self.engine.set_params(net, flat, "ema")
