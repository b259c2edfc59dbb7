"""HipEngine -- thin Python wrapper over the C ABI (include/rsrgan.h).

PyTorch is only plumbing here: it owns the device buffers that are passed to the library as raw
pointers (Tensor.data_ptr()) and the HIP stream the library enqueues on.  No arithmetic of the
GAN step happens in torch."""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import NET_D, NET_G, RsrganCfg, SCALARS, WHAT, check


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class _RawDeviceBuffer:
    """Exposes a device pointer owned by librsrgan_hip through __cuda_array_interface__ so that
    torch can alias it (zero copy) -- needed to all-reduce the gradient buffer with RCCL."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr, False),
                                         "version": 2, "strides": None}


class HipEngine:
    def __init__(self, *, batch_size: int, max_frames: int, input_dim: int = 257, output_dim: int = 40,
                 g_type: str = "lstm", g_layers: Optional[int] = None, g_cells: Optional[int] = None,
                 g_proj: Optional[int] = None, d_layers: Optional[int] = None, d_cells: Optional[int] = None,
                 d_proj: Optional[int] = None, d_type: Optional[str] = None, d_joint_off: Optional[int] = None,
                 d_joint_dim: Optional[int] = None, clip_norm: Optional[float] = None, g_splice: Optional[int] = None,
                 l2_scale: float = 0.0, cross_validation: bool = False, batch_norm: bool = False,
                 ema_decay: float = 0.9999, seed: int = 4321, device: Optional[torch.device] = None, flags: int = 3):
        if g_type not in _lib.G_TYPES:
            raise ValueError("Unrecognized G type {}".format(g_type))      # gan_rnn_placeholder.py:131-132
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.RsrganError("no GPU visible: rsrgan_amd runs only on MI355X (gfx950); there is no CPU fallback")
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        torch.cuda.set_device(self.device)
        cfg = RsrganCfg()
        check(self.lib.rsrgan_default_cfg(_lib.G_TYPES[g_type], C.byref(cfg)))
        cfg.batch_size, cfg.max_frames, cfg.input_dim, cfg.output_dim = batch_size, max_frames, input_dim, output_dim
        for k, v in dict(g_layers=g_layers, g_cells=g_cells, g_proj=g_proj, d_layers=d_layers, d_cells=d_cells,
                         d_proj=d_proj).items():
            if v is not None:
                setattr(cfg, k, int(v))
        if d_type is not None:
            if d_type not in _lib.D_TYPES:
                raise ValueError("Unrecognized D type {}".format(d_type))
            cfg.d_type = _lib.D_TYPES[d_type]
            if d_type == "dnn" and g_type not in ("dnn", "rced"):          # discriminator_dnn on the 40-dim target only
                cfg.d_layers = d_layers or 4
                cfg.d_cells = d_cells or 1024
                cfg.d_joint_off, cfg.d_joint_dim = 0, 0
        if d_joint_off is not None:
            cfg.d_joint_off = int(d_joint_off)
        if d_joint_dim is not None:
            cfg.d_joint_dim = int(d_joint_dim)
        if clip_norm is not None:
            cfg.clip_norm = float(clip_norm)
        if g_splice is not None:
            cfg.g_splice = int(g_splice)
        cfg.l2_scale = l2_scale
        cfg.cross_validation = 1 if cross_validation else 0
        cfg.ema_decay = ema_decay
        cfg.flags = flags | (_lib.FLAG_BATCH_NORM if batch_norm else 0)
        self.cfg = cfg
        self.batch_size, self.max_frames = batch_size, max_frames
        self.input_dim, self.output_dim = input_dim, output_dim
        self.h = C.c_void_p()
        # RSRGAN_DPIPE (include/rsrgan.h rsrgan_d_step): the library may read a D-run's labels and lengths ahead of the stream if the
        # caller vouches that they are complete when the call is made.  THIS layer can always vouch: d_backward hands every labels /
        # lengths tensor through upload_ready (copied or converted on the upload stream, host-waited), so the pipelined D-run is the
        # default of the Python mirror -- and what bench.py times.  The C ABI's own default stays off: a raw caller has to opt in.
        os.environ.setdefault("RSRGAN_DPIPE", "1")
        check(self.lib.rsrgan_create(C.byref(cfg), C.c_uint64(seed), C.byref(self.h)))
        self._grad_views = {}
        self._comm_stream = None
        # hipGraph capture needs a real stream: the library moves work handed to the legacy null stream onto its own stream
        # with an event hop on both sides of EVERY call (measured ~0.2 ms per call); loops that run under on_stream() hand it
        # this stream instead and pay nothing
        self.stream = torch.cuda.Stream(device=self.device)
        self.d_has_adam = g_type in ("dnn", "rced")
        self.ema_enabled = ema_decay > 0          # no EMA shadow buffers in the library otherwise (WHAT['ema'] is absent)

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.rsrgan_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing ----------------------------------------------------------------------
    @contextlib.contextmanager
    def on_stream(self):
        """Make the engine's stream the current torch stream for the block (ordered after the previous current stream on
        entry, and the previous stream after it on exit).  Nested use is free."""
        prev = torch.cuda.current_stream(self.device)
        if prev == self.stream:
            yield
            return
        self.stream.wait_stream(prev)
        try:
            with torch.cuda.stream(self.stream):
                yield
        finally:                       # also when the block raises: later work on `prev` must still see what was queued here
            prev.wait_stream(self.stream)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _f32(self, a, shape=None) -> torch.Tensor:
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        t = t.to(self.device, torch.float32, non_blocking=True).contiguous()
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError("expected shape %s, got %s" % (tuple(shape), tuple(t.shape)))
        return t

    def upload_ready(self, a, int32=False) -> torch.Tensor:
        """Host array -> device tensor that is COMPLETE when this returns (copied on an upload stream of its own, which the host
        waits for -- not for the compute stream): what RSRGAN_DPIPE=1 asks of the labels and lengths of a D-run (include/rsrgan.h
        rsrgan_d_step), so that D(real) of the next step can run beside the previous step's tail.  Device tensors are converted on the
        upload stream if they need it, and waited for once per tensor object otherwise (see below)."""
        us = getattr(self, "_upload_stream", None)
        if us is None:
            us = self._upload_stream = torch.cuda.Stream(self.device)
        if isinstance(a, torch.Tensor) and a.device == self.device:
            want = torch.int32 if int32 else torch.float32
            if a.dtype == want and a.is_contiguous():
                # A device tensor nobody has vouched for may still be being written by a kernel queued on some stream.  The first time
                # this very tensor object (at this version: in-place writes through torch bump it) comes by, the host waits for the
                # device once; from then on it passes through untouched -- a resident batch fed again and again (bench.py, a cached
                # validation set) costs nothing, a freshly computed tensor per step costs a device wait per step (or RSRGAN_DPIPE=0).
                seen = self.__dict__.setdefault("_vouched", {})
                ent = seen.get(id(a))
                if ent is None or ent[0]() is not a or ent[1] != a._version:
                    torch.cuda.synchronize(self.device)
                    self._vouch(a)
                return a
            # an int64 `lengths`, a float64 or strided label: the conversion is a KERNEL.  Queued on the current stream it would sit
            # behind that stream's backlog while D(real) on the library's side stream reads its output ahead of it -- so it runs on
            # the upload stream (the caller vouches for `a` itself being complete) and the host waits for it, as for a host array.
            with torch.cuda.stream(us):
                t = a.to(want).contiguous()
            us.synchronize()
            t.record_stream(torch.cuda.current_stream(self.device))
            self._vouch(t)
            return t
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a) if int32 else np.ascontiguousarray(a, dtype=np.float32))
        with torch.cuda.stream(us):
            t = t.to(self.device, non_blocking=True)
            t = (t.to(torch.int32) if int32 else t.to(torch.float32)).contiguous()
        us.synchronize()
        t.record_stream(torch.cuda.current_stream(self.device))
        self._vouch(t)
        return t

    def _vouch(self, t):
        """remember that device tensor `t` (this object, at this version) is complete: upload_ready lets it pass without a wait"""
        import weakref
        seen = self.__dict__.setdefault("_vouched", {})
        if len(seen) > 256:
            for k in [k for k, (r, _) in seen.items() if r() is None]:
                del seen[k]
            if len(seen) > 256:
                seen.clear()
        seen[id(t)] = (weakref.ref(t), t._version)

    def upload_async(self, a, int32=False) -> torch.Tensor:
        """Host array -> device tensor on the upload stream, consumed in STREAM order: the current stream waits for the copy (an event,
        no host wait), so the copy runs beside whatever the compute stream still has queued instead of in front of the next step's
        first launch (a 6.6 MB pageable copy of the input frames is ~0.6 ms of a 4.8 ms step).  What the inputs of every run, and the
        labels / lengths of a stream-ordered D-run (RSRGAN_DPIPE=0), need; device tensors pass through."""
        if isinstance(a, torch.Tensor) and a.device == self.device:
            return a
        us = getattr(self, "_upload_stream", None)
        if us is None:
            us = self._upload_stream = torch.cuda.Stream(self.device)
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a) if int32 else np.ascontiguousarray(a, dtype=np.float32))
        with torch.cuda.stream(us):
            t = t.to(self.device, non_blocking=True)
            t = (t.to(torch.int32) if int32 else t.to(torch.float32)).contiguous()
            ev = torch.cuda.Event()
            ev.record(us)
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        t.record_stream(cur)
        return t

    def _i32(self, a) -> torch.Tensor:
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        return t.to(self.device).to(torch.int32).contiguous()       # placeholder is float32, cast like dynamic_rnn

    def _noise(self, n):
        if n is None:
            return None
        return self._f32(n).reshape(self.batch_size, self.output_dim)

    # -- scalars -----------------------------------------------------------------------
    def set_scalar(self, name: str, v: float):
        check(self.lib.rsrgan_set_scalar(self.h, SCALARS[name], float(v)))

    def get_scalar(self, name: str) -> float:
        out = C.c_double()
        check(self.lib.rsrgan_get_scalar(self.h, SCALARS[name], C.byref(out)))
        return out.value

    # -- variables ---------------------------------------------------------------------
    def tensor_table(self, net: int) -> List[Tuple[str, Tuple[int, ...], int]]:
        n = self.lib.rsrgan_num_tensors(self.h, net)
        out = []
        buf = C.create_string_buffer(256)
        for i in range(n):
            r, c, off = C.c_int32(), C.c_int32(), C.c_int64()
            check(self.lib.rsrgan_tensor_info(self.h, net, i, buf, 256, C.byref(r), C.byref(c), C.byref(off)))
            shape = (r.value,) if c.value == 0 else (r.value, c.value)
            out.append((buf.value.decode(), shape, off.value))
        return out

    def param_count(self, net: int) -> int:
        return int(self.lib.rsrgan_param_count(self.h, net))

    def get_params(self, net: int, what: str = "variables") -> torch.Tensor:
        out = torch.empty(self.param_count(net), dtype=torch.float32, device=self.device)
        check(self.lib.rsrgan_get_params(self.h, net, WHAT[what], _ptr(out), self._stream()))
        return out

    def set_params(self, net: int, flat, what: str = "variables"):
        t = self._f32(flat).reshape(-1)
        if t.numel() != self.param_count(net):
            raise ValueError("expected %d floats, got %d" % (self.param_count(net), t.numel()))
        check(self.lib.rsrgan_set_params(self.h, net, WHAT[what], _ptr(t), self._stream()))
        torch.cuda.current_stream(self.device).synchronize()        # `t` may be a temporary

    def get_grads(self, net: int) -> torch.Tensor:
        out = torch.empty(self.param_count(net), dtype=torch.float32, device=self.device)
        check(self.lib.rsrgan_get_grads(self.h, net, _ptr(out), self._stream()))
        return out

    def grad_view(self, net: int) -> torch.Tensor:
        """Zero-copy torch view of the library's (padded, flat) gradient buffer."""
        if net not in self._grad_views:
            p, n = C.c_void_p(), C.c_int64()
            check(self.lib.rsrgan_grad_buffer(self.h, net, C.byref(p), C.byref(n)))
            self._grad_views[net] = torch.as_tensor(_RawDeviceBuffer(p.value, n.value), device=self.device)
        return self._grad_views[net]

    def grad_buckets(self, net: int) -> List[Tuple[int, int]]:
        """(offset, count) float ranges of the gradient buffer in the order the backward completes them."""
        out = []
        for i in range(self.lib.rsrgan_grad_bucket_count(self.h, net)):
            off, cnt = C.c_int64(), C.c_int64()
            check(self.lib.rsrgan_grad_bucket_info(self.h, net, i, C.byref(off), C.byref(cnt)))
            out.append((off.value, cnt.value))
        return out

    def all_reduce_grads(self, net: int, group=None, force: bool = False):
        """average_gradients (utils/ops.py:343-376) over ranks: one all-reduce per gradient bucket, issued on a
        communication stream that waits only for that bucket's completion event, so RCCL moves the first buckets over
        xGMI while the weight-gradient GEMMs of the later ones still run.  RSRGAN_BUCKETED_ALLREDUCE=0 (or a single
        bucket) falls back to one all-reduce of the whole buffer on the compute stream."""
        from . import dist as rdist
        ws = rdist.world_size(group)
        if ws == 1 and not force:
            return
        view = self.grad_view(net)
        buckets = self.grad_buckets(net)
        if len(buckets) <= 1 or os.environ.get("RSRGAN_BUCKETED_ALLREDUCE", "1") == "0":
            rdist.all_reduce_mean_(view, group)
            return
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=self.device)

        def wait_bucket(i, stream):
            check(self.lib.rsrgan_grad_bucket_wait(self.h, net, i, C.c_void_p(stream.cuda_stream)))

        timing = {"buckets": []} if getattr(self, "bucket_timing", False) else None
        rdist.all_reduce_mean_buckets_(view, buckets, group, wait_bucket, self._comm_stream, timing)   # joins before rsrgan_apply
        if timing is not None:
            self._last_timing = timing

    def bucket_report(self):
        """Diagnostic of the last bucketed all-reduce run with `engine.bucket_timing = True` (synchronises): per bucket its bytes,
        the milliseconds its all-reduce took on the communication stream and when it started / ended relative to the moment the
        compute stream had finished every gradient; `exposed_ms` = how long the compute stream then waited for the communication
        stream (everything before that moment was hidden behind the weight-gradient GEMMs of the later buckets)."""
        t = getattr(self, "_last_timing", None)
        if not t:
            return None
        torch.cuda.synchronize(self.device)
        rows = []
        for i, nbytes, e0, e1 in t["buckets"]:
            rows.append({"bucket": i, "bytes": int(nbytes), "allreduce_ms": round(e0.elapsed_time(e1), 4),
                         "start_vs_compute_done_ms": round(t["ready"].elapsed_time(e0), 4),
                         "end_vs_compute_done_ms": round(t["ready"].elapsed_time(e1), 4)})
        return {"buckets": rows, "exposed_ms": round(t["ready"].elapsed_time(t["joined"]), 4)}

    # -- the path ----------------------------------------------------------------------
    def _check_batch(self, x, lab=None, ln=None):
        """The library reads and writes exactly batch_size x T x input_dim / output_dim elements behind the raw pointers:
        refuse anything else here (the reference's placeholders have these static shapes, gan_rnn_placeholder.py:94-104)."""
        if x.dim() != 3 or x.shape[0] != self.batch_size or x.shape[2] != self.input_dim:
            raise ValueError("inputs must be [batch_size=%d, T, input_dim=%d], got %s" % (self.batch_size, self.input_dim, tuple(x.shape)))
        T = x.shape[1]
        if not 0 < T <= self.max_frames:
            raise ValueError("T=%d outside (0, max_frames=%d]" % (T, self.max_frames))
        if lab is not None and tuple(lab.shape) != (self.batch_size, T, self.output_dim):
            raise ValueError("labels must be [%d, %d, %d], got %s" % (self.batch_size, T, self.output_dim, tuple(lab.shape)))
        if ln is not None and ln.numel() != self.batch_size:
            raise ValueError("lengths must hold batch_size=%d entries, got %d" % (self.batch_size, ln.numel()))
        return T

    def forward_g(self, x, lengths) -> torch.Tensor:
        x = self._f32(x)
        ln = self._i32(lengths) if lengths is not None else None
        T = self._check_batch(x, None, ln)
        B = self.batch_size
        y = torch.empty(B, T, self.output_dim, dtype=torch.float32, device=self.device)
        check(self.lib.rsrgan_forward_g(self.h, _ptr(x), _ptr(ln), T, _ptr(y), self._stream()))
        return y

    def d_backward(self, x, lab, lengths, noise_real=None, noise_fake=None, train=True, apply=False) -> torch.Tensor:
        if os.environ.get("RSRGAN_DPIPE", "0") not in ("", "0"):      # the library reads labels / lengths ahead of the stream: hand it complete ones
            lab = self.upload_ready(lab)
            lengths = self.upload_ready(lengths, int32=True) if lengths is not None else None
        x, lab = self._f32(x), self._f32(lab)
        ln = self._i32(lengths) if lengths is not None else None
        nr, nf = self._noise(noise_real), self._noise(noise_fake)
        out = torch.empty(3, dtype=torch.float32, device=self.device)
        T = self._check_batch(x, lab, ln)
        if apply or not train:
            check(self.lib.rsrgan_d_step(self.h, _ptr(x), _ptr(lab), _ptr(ln), T, _ptr(nr), _ptr(nf), _ptr(out),
                                         1 if train else 0, self._stream()))
        else:
            check(self.lib.rsrgan_d_backward(self.h, _ptr(x), _ptr(lab), _ptr(ln), T, _ptr(nr), _ptr(nf), _ptr(out),
                                             self._stream()))
        self._keep = (x, lab, ln, nr, nf)       # borrowed until the stream has consumed them
        return out

    def g_backward(self, x, lab, lengths, noise_fake=None, train=True, reuse=False, apply=False) -> torch.Tensor:
        x, lab = self._f32(x), self._f32(lab)
        ln = self._i32(lengths) if lengths is not None else None
        nf = self._noise(noise_fake)
        out = torch.empty(4, dtype=torch.float32, device=self.device)
        T = self._check_batch(x, lab, ln)
        if apply or not train:
            check(self.lib.rsrgan_g_step(self.h, _ptr(x), _ptr(lab), _ptr(ln), T, _ptr(nf), _ptr(out),
                                         1 if train else 0, 1 if reuse else 0, self._stream()))
        else:
            check(self.lib.rsrgan_g_backward(self.h, _ptr(x), _ptr(lab), _ptr(ln), T, _ptr(nf), _ptr(out),
                                             1 if reuse else 0, self._stream()))
        self._keep2 = (x, lab, ln, nf)
        return out

    def apply(self, net: int):
        check(self.lib.rsrgan_apply(self.h, net, self._stream()))

    def profile_begin(self):
        check(self.lib.rsrgan_profile_begin(self.h))

    def profile_read_kind(self, kind):
        """(launches, total_us, algorithmic_flops) of kernel class `kind` (1 = k_glstm_fwd) since profile_begin; call before profile_read"""
        n, us, fl = C.c_int32(), C.c_double(), C.c_double()
        check(self.lib.rsrgan_profile_read_kind(self.h, kind, C.byref(n), C.byref(us), C.byref(fl)))
        return n.value, us.value, fl.value

    def profile_read(self):
        """(launches, total_us, algorithmic_flops) of k_fwd_gates since profile_begin (synchronises)."""
        n, us, fl = C.c_int32(), C.c_double(), C.c_double()
        check(self.lib.rsrgan_profile_read(self.h, C.byref(n), C.byref(us), C.byref(fl)))
        return n.value, us.value, fl.value

    def device_status(self):
        """0, or 1 + the first workgroup of a persistent recurrence launch whose bounded wait expired (synchronises; clears the word)"""
        code = C.c_int32()
        check(self.lib.rsrgan_device_status(self.h, C.byref(code)))
        return code.value

    def set_dropout(self, keep_prob, seed=0):
        """tf.nn.dropout(h, keep_prob) after every hidden ReLU of the frame-level nets (dnn.py:116-121); `seed` = mask stream"""
        check(self.lib.rsrgan_set_dropout(self.h, float(keep_prob), int(seed) & 0xFFFFFFFFFFFFFFFF))

    def profile_launches(self):
        """launches of the recurrence kernels the host issued since profile_begin (the profiled step runs eagerly)"""
        n = C.c_int64()
        check(self.lib.rsrgan_profile_launches(self.h, C.byref(n)))
        return n.value

    def launch_floor(self, n=400, mode=1):
        """us per dependent launch of a replayed graph of n launches (mode 0: empty kernels, 1: one dependent operand round trip)"""
        us = C.c_double()
        check(self.lib.rsrgan_op_launch_floor(n, mode, C.byref(us), self._stream()))
        return us.value

    # -- low-level op (unit tests, micro-bench) ------------------------------------------
    def op_gemm(self, A, a_kc, B, b_kc, C_, M, N, K, bias=None, act=0, alpha=0.3, accumulate=False):
        check(self.lib.rsrgan_op_gemm(_ptr(A), A.stride(0), 1 if a_kc else 0, _ptr(B), B.stride(0), 1 if b_kc else 0,
                                      _ptr(C_), C_.stride(0), M, N, K, _ptr(bias), act, alpha, 1 if accumulate else 0,
                                      self._stream()))
