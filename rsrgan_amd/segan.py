"""SEGAN -- host-side mirror of models/segan.py:SEGAN (AEGenerator + conv discriminator with virtual batch norm, LSGAN + L1,
RMSProp for both nets; BASELINE.json configs[4]) on the C ABI `rsrgan_segan_*` of include/rsrgan.h.

The reference's trainer cannot run as shipped (models/segan.py:136 calls an undefined variables_on_gpu0(); scripts/train_segan.py:20
imports a missing utils.utils -- SURVEY 0-D7), but the graph it would build is fully specified; this class exposes what
scripts/train_segan.py:24-82 drives:  sess.run([model.d_opt, model.d_losses[0]]) -> d_step(),  sess.run([model.g_opt,
model.g_losses[0]]) -> g_step(),  model.Gs -> forward(), save/load, the mutable scalars.  The reference bakes the batch tensors and
three random draws into the graph (z: generator.py:201-205; one gaussian_noise_layer draw per discriminator call:
discriminator.py:74); here a batch is passed to each step, and the draws are made on the device per call unless the caller
injects them (parity tests).  Every tower of the reference sees the SAME batch (segan.py:134-137: no slicing); with
process_group set the ranks are the towers -- each rank passes its own batch, gradients are averaged by an RCCL all-reduce
(utils/ops.py:343-376) before the update."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from . import dist as rdist
from ._lib import NET_D, NET_G, check

DEPTHS = (16, 32, 32, 64, 64, 128, 128, 256, 256, 512, 1024)            # models/segan.py:89,91


class _Raw:
    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 2, "strides": None}


class SEGAN(object):
    def __init__(self, sess, args, devices, inputs=None, labels=None, cross_validation=False, name="SEGAN", *, process_group=None,
                 seed: int = 4321, depths: Optional[Tuple[int, ...]] = None, g_kwidth: int = 20, d_kwidth: int = 31):
        self.name, self.sess, self.cross_validation = name, sess, cross_validation
        if getattr(args, "g_type", "ae") != "ae":
            raise ValueError("Unrecognized G type {}".format(args.g_type))                  # segan.py:117-118 ('dfeat' is not built)
        if getattr(args, "deconv_type", "deconv") != "deconv":
            raise ValueError("Unknown deconv type {}".format(args.deconv_type))            # generator.py:244 ('nn_deconv' is not built)
        for flag in ("bias_downconv", "bias_deconv", "bias_D_conv"):
            if not getattr(args, flag, True):
                raise NotImplementedError("%s=False: the shipped recipe biases every conv (run_segan.sh:114-116)" % flag)
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.RsrganError("no GPU visible: rsrgan_amd runs only on MI355X (gfx950); there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.batch_size = int(args.batch_size)
        self.input_len = int(args.input_dim) * (int(getattr(args, "left_context", 0)) + 1 + int(getattr(args, "right_context", 0)))
        self.output_dim = int(args.output_dim)
        self.save_dir = getattr(args, "save_dir", None)
        self.disc_updates, self.d_clip_weights = 1, False                                # segan.py:79,84
        self.init_noise_std = float(getattr(args, "init_noise_std", 0.0))
        self.disc_noise_std = self.init_noise_std                                        # tf.Variable, segan.py:93
        depths = tuple(depths or DEPTHS)
        cfg = _lib.SeganCfg()
        check(self.lib.rsrgan_segan_default_cfg(C.byref(cfg)))
        cfg.batch_size, cfg.input_len, cfg.output_dim, cfg.n_layers = self.batch_size, self.input_len, self.output_dim, len(depths)
        for i, d in enumerate(depths):
            cfg.g_depths[i] = d
            cfg.d_depths[i] = d
        cfg.g_kwidth, cfg.d_kwidth = g_kwidth, d_kwidth
        cfg.g_prelu = 1 if getattr(args, "g_nl", "prelu") == "prelu" else 0
        self.cfg, self.depths = cfg, depths
        self.h = C.c_void_p()
        check(self.lib.rsrgan_segan_create(C.byref(cfg), C.c_uint64(seed), C.byref(self.h)))
        self.code_len = self.input_len
        for _ in depths:
            self.code_len = (self.code_len + 1) // 2
        # (process_group=None under torchrun = the default group, as in GAN / GAN_RNN: the towers' gradients are averaged)
        self.group, self.world = process_group, rdist.world_size(process_group)
        self.set_scalar("g_learning_rate", float(getattr(args, "g_learning_rate", 1e-3)))
        self.set_scalar("d_learning_rate", float(getattr(args, "d_learning_rate", 1e-3)))
        self.set_scalar("l1_lambda", float(getattr(args, "init_l1_weight", 100.0)))
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(seed + 17)

    _S = {"g_learning_rate": 0, "d_learning_rate": 1, "l1_lambda": 2}

    def set_scalar(self, name, v):
        """sess.run(tf.assign(model.<scalar>, v))"""
        if name == "disc_noise_std":
            self.disc_noise_std = float(v)
            return
        check(self.lib.rsrgan_segan_set_scalar(self.h, self._S[name], C.c_double(float(v))))
        setattr(self, name, float(v))

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.rsrgan_segan_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- plumbing
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _f32(self, a, shape):
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        t = t.to(self.device, torch.float32).contiguous()
        if tuple(t.shape) != tuple(shape):
            raise ValueError("expected shape %s, got %s" % (tuple(shape), tuple(t.shape)))
        return t

    def _draws(self, z, noises, n):
        B, Lj = self.batch_size, self.input_len + self.output_dim
        if z is None:
            z = torch.randn(B, self.code_len, self.depths[-1], device=self.device, generator=self._gen)
        z = self._f32(z, (B, self.code_len, self.depths[-1]))
        out = []
        for i in range(n):
            nz = noises[i] if noises is not None else None
            if nz is None and self.disc_noise_std > 0:
                nz = self.disc_noise_std * torch.randn(B, Lj, device=self.device, generator=self._gen)
            out.append(None if nz is None else self._f32(nz, (B, Lj)))
        return z, out

    @staticmethod
    def _p(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def tensor_table(self, net) -> List[Tuple[str, Tuple[int, ...], int]]:
        out = []
        name = C.create_string_buffer(160)
        for i in range(self.lib.rsrgan_segan_num_tensors(self.h, net)):
            r, c, off = C.c_int32(), C.c_int32(), C.c_int64()
            check(self.lib.rsrgan_segan_tensor_info(self.h, net, i, name, 160, C.byref(r), C.byref(c), C.byref(off)))
            out.append((name.value.decode(), (r.value,) if c.value == 0 else (r.value, c.value), off.value))
        return out

    def _tf_shape(self, name, shape):
        """the variable's TensorFlow shape from the library's 2-D one"""
        k = self.cfg.g_kwidth if name.startswith("g_") else self.cfg.d_kwidth
        if name.endswith("/W") and "logits_conv" not in name:
            return (k, 1, shape[0] // k, shape[1])
        if "logits_conv" in name:
            return (k, shape[0] // k, 1)
        return tuple(shape)

    def _flat(self, net, what):
        n = self.lib.rsrgan_segan_param_count(self.h, net)
        t = torch.empty(n, dtype=torch.float32, device=self.device)
        check(self.lib.rsrgan_segan_get_params(self.h, net, what, self._p(t), self._stream()))
        return t

    def get_vars(self, what=0) -> Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray]]:
        out = []
        for net in (NET_G, NET_D):
            flat = self._flat(net, what).cpu().numpy()
            out.append({nm: flat[off:off + int(np.prod(sh))].reshape(self._tf_shape(nm, sh)) for nm, sh, off in self.tensor_table(net)})
        return out[0], out[1]

    def get_grads(self, net):
        flat = self._flat(net, 2).cpu().numpy()
        return {nm: flat[off:off + int(np.prod(sh))].reshape(self._tf_shape(nm, sh)) for nm, sh, off in self.tensor_table(net)}

    def set_vars(self, g=None, d=None, what=0):
        for net, vals in ((NET_G, g), (NET_D, d)):
            if vals is None:
                continue
            flat = np.concatenate([np.asarray(vals[nm], np.float32).reshape(-1) for nm, _, _ in self.tensor_table(net)])
            t = torch.from_numpy(flat).to(self.device)
            check(self.lib.rsrgan_segan_set_params(self.h, net, what, self._p(t), self._stream()))
            torch.cuda.current_stream(self.device).synchronize()

    def _all_reduce(self, net):
        if self.world <= 1:
            return
        ptr, cnt = C.c_void_p(), C.c_int64()
        check(self.lib.rsrgan_segan_grad_buffer(self.h, net, C.byref(ptr), C.byref(cnt)))
        rdist.all_reduce_mean_(torch.as_tensor(_Raw(ptr.value, cnt.value), device=self.device), self.group)

    # ---- the fetches of scripts/train_segan.py
    def forward(self, inputs, z=None):
        """model.Gs[0] (segan.py:194-197): G(inputs) [B, output_dim]"""
        x = self._f32(inputs, (self.batch_size, self.input_len))
        z, _ = self._draws(z, None, 0)
        y = torch.empty(self.batch_size, self.output_dim, dtype=torch.float32, device=self.device)
        check(self.lib.rsrgan_segan_forward_g(self.h, self._p(x), self._p(z), self._p(y), self._stream()))
        return y.cpu().numpy()

    def d_step(self, inputs, labels, z=None, noises=None, train=True, apply=True):
        """sess.run([model.d_opt, model.d_losses[0]]) (train_segan.py:32-36) -> (d_rl_loss, d_fk_loss, d_loss).  noises = (reference
        pass, real, fake) gaussian_noise_layer draws or None; train=False: the eval fetch (:68)."""
        x = self._f32(inputs, (self.batch_size, self.input_len)); lab = self._f32(labels, (self.batch_size, self.output_dim))
        z, nz = self._draws(z, noises, 3)
        out = torch.empty(3, dtype=torch.float32, device=self.device)
        check(self.lib.rsrgan_segan_d_backward(self.h, self._p(x), self._p(lab), self._p(z), self._p(nz[0]), self._p(nz[1]), self._p(nz[2]),
                                               self._p(out), 1 if train else 0, self._stream()))
        if train and apply:
            self._all_reduce(NET_D)
            check(self.lib.rsrgan_segan_apply(self.h, NET_D, self._stream()))
        return out.cpu().numpy()

    def g_step(self, inputs, labels, z=None, noises=None, train=True, apply=True):
        """sess.run([model.g_opt, model.g_losses[0]]) (train_segan.py:40-44) -> (g_adv_loss, g_l1_loss, g_loss).  noises = (reference
        pass, fake)."""
        x = self._f32(inputs, (self.batch_size, self.input_len)); lab = self._f32(labels, (self.batch_size, self.output_dim))
        z, nz = self._draws(z, noises, 2)
        out = torch.empty(3, dtype=torch.float32, device=self.device)
        check(self.lib.rsrgan_segan_g_backward(self.h, self._p(x), self._p(lab), self._p(z), self._p(nz[0]), self._p(nz[1]), self._p(out),
                                               1 if train else 0, self._stream()))
        if train and apply:
            self._all_reduce(NET_G)
            check(self.lib.rsrgan_segan_apply(self.h, NET_G, self._stream()))
        return out.cpu().numpy()

    # ---- tf.train.Saver (segan.py:26-54): variables + RMSProp slots
    def save(self, save_dir, step):
        """rank 0 writes; every rank returns once the checkpoint exists (or raises if rank 0 could not write it)"""
        os.makedirs(save_dir, exist_ok=True)
        return rdist.run_on_rank0(lambda: self._save_rank0(save_dir, step), self.group)

    def _save_rank0(self, save_dir, step):
        g, d = self.get_vars(0)
        gm, dm = self.get_vars(1)
        base = "%s-%d" % (self.name, int(step))
        path = os.path.join(save_dir, base + ".npz")
        np.savez(path, **{"v/" + k: v for k, v in {**g, **d}.items()}, **{"rms/" + k: v for k, v in {**gm, **dm}.items()})
        with open(os.path.join(save_dir, "checkpoint"), "w") as f:          # the state file of Model.save (gan_rnn.py)
            f.write('model_checkpoint_path: "%s"\n' % base)
        return path

    def load(self, save_dir, model_file=None):
        if not save_dir or not os.path.exists(save_dir):
            print("[!] Checkpoints path does not exist...")
            return False
        if model_file is None:
            ck = os.path.join(save_dir, "checkpoint")
            if not os.path.exists(ck):
                return False
            with open(ck) as f:
                line = f.readline().strip()
            model_file = line.split('"')[1] if '"' in line else line        # (a bare file name: checkpoints written before round 4)
        if not model_file.endswith(".npz"):
            model_file += ".npz"
        data = np.load(os.path.join(save_dir, model_file))
        for what, pre in ((0, "v/"), (1, "rms/")):
            g = {nm: data[pre + nm] for nm, _, _ in self.tensor_table(NET_G)}
            d = {nm: data[pre + nm] for nm, _, _ in self.tensor_table(NET_D)}
            self.set_vars(g, d, what)
        return True
