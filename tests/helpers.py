"""Shared test helpers: seeded synthetic data, an oracle-backed engine for the CPU tests of the
host logic (multi-process gloo included), and builders that pair the HIP model with the oracle."""
from types import SimpleNamespace

import numpy as np
import torch

from oracle import rsrgan_oracle as O

NET_G, NET_D = 0, 1


def small_cfg(g_type="lstm", **kw):
    c = O.NetCfg(input_dim=9, output_dim=5, g_type=g_type, g_layers=2, g_cells=12, g_proj=7,
                 d_layers=2, d_cells=8, d_proj=5)
    if g_type != "lstm":
        c.g_proj = 9 if g_type == "res_lstm_l" else 7
        c.g_layers = 3
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def rand_params(cfg, seed=0, bias_std=0.1, dtype=np.float32):
    """xavier-uniform weights (+ small random biases so every gradient path is exercised),
    rounded to fp32 so HIP and oracle start from bit-identical values."""
    rng = np.random.default_rng(seed)
    g = O.xavier_init(O.g_param_specs(cfg), rng)
    d = O.xavier_init(O.d_param_specs(cfg), rng)
    for p in (g, d):
        for k in p:
            if "bias" in k and bias_std > 0:
                p[k] = rng.normal(0, bias_std, p[k].shape)
            p[k] = p[k].astype(dtype)
    return g, d


def rand_batch(cfg, B, T, seed=1234, ragged=False):
    """SURVEY 8d synthetic inputs: inputs, labels ~ N(0,1) f32, lengths = T (or ~U{T/2..T})."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, T, cfg.input_dim)).astype(np.float32)
    lab = rng.standard_normal((B, T, cfg.output_dim)).astype(np.float32)
    ln = rng.integers(max(T // 2, 1), T + 1, size=B).astype(np.int32) if ragged else np.full(B, T, np.int32)
    if ragged:
        ln[0] = T
    return x, lab, ln


def args_for(cfg, B, **kw):
    a = SimpleNamespace(batch_size=B, input_dim=cfg.input_dim, output_dim=cfg.output_dim, left_context=0,
                        right_context=0, g_type=cfg.g_type, keep_prob=1.0, batch_norm=False, num_gpu=1,
                        save_dir=None, l2_scale=0.0, disc_updates=1, gen_updates=1, init_mse_weight=10.0,
                        init_disc_noise_std=0.0, d_learning_rate=1e-3, g_learning_rate=8e-5)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def overrides(cfg):
    o = dict(g_layers=cfg.g_layers, g_cells=cfg.g_cells, g_proj=cfg.g_proj, d_layers=cfg.d_layers,
             d_cells=cfg.d_cells, d_proj=cfg.d_proj)
    if getattr(cfg, "d_type", "lstm") != "lstm":
        o["d_type"] = cfg.d_type
    return o


def split_flat(flat, table):
    out = {}
    for name, shape, off in table:
        n = int(np.prod(shape))
        out[name] = np.asarray(flat[off:off + n]).reshape(shape)
    return out


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


class OracleEngine:
    """Implements the engine protocol GAN_RNN expects (see rsrgan_amd/engine_hip.py) with the numpy
    oracle, on CPU tensors.  TEST ONLY: lets the host logic (sharding, all-reduce order, scalar
    plumbing, checkpoints, train_one_iteration) run under `-m "not gpu"` and under gloo."""
    ema_enabled = True

    def __init__(self, cfg, g, d, batch_size, l2_scale=0.0, cross_validation=False, dtype=np.float64):
        self.cfg = cfg
        self.device = torch.device("cpu")
        self.batch_size = batch_size
        self.output_dim = cfg.output_dim
        self.o = O.GanRnnOracle(cfg, g, d, batch_size=batch_size, l2_scale=l2_scale,
                                cross_validation=cross_validation, dtype=dtype)
        self.specs = {NET_G: O.g_param_specs(cfg), NET_D: O.d_param_specs(cfg)}
        self._grads = {n: torch.zeros(self.param_count(n), dtype=torch.float64) for n in (NET_G, NET_D)}
        self._cache = None

    def tensor_table(self, net):
        out, off = [], 0
        for name, shape in self.specs[net]:
            out.append((name, tuple(shape), off))
            off += int(np.prod(shape))
        return out

    def param_count(self, net):
        return sum(int(np.prod(s)) for _, s in self.specs[net])

    def _dict(self, net, what):
        o = self.o
        return {(NET_G, "variables"): o.g, (NET_D, "variables"): o.d, (NET_G, "adam_m"): o.adam_m,
                (NET_G, "adam_v"): o.adam_v, (NET_G, "ema"): o.g_ema, (NET_D, "ema"): o.d_ema}[(net, what)]

    def get_params(self, net, what="variables"):
        d = self._dict(net, what)
        return torch.from_numpy(np.concatenate([d[n].reshape(-1) for n, _ in self.specs[net]]).astype(np.float32))

    def set_params(self, net, flat, what="variables"):
        d = self._dict(net, what)
        flat = np.asarray(flat, np.float64).reshape(-1)
        for name, shape, off in self.tensor_table(net):
            d[name] = flat[off:off + int(np.prod(shape))].reshape(shape).copy()

    _S = {"g_learning_rate": "g_learning_rate", "d_learning_rate": "d_learning_rate", "mse_lambda": "mse_lambda",
          "d_real": "d_real", "d_fake": "d_fake", "l2_scale": "l2_scale", "clip_norm": "clip_norm", "adam_step": "adam_t"}

    def set_scalar(self, name, v):
        setattr(self.o, self._S[name], int(v) if name == "adam_step" else float(np.float32(v)))

    def get_scalar(self, name):
        return float(getattr(self.o, self._S[name]))

    def grad_view(self, net):
        return self._grads[net]

    bucketed = False            # True: expose the HipEngine bucket protocol (per-tensor ranges, last tensor first)

    def grad_buckets(self, net):
        tt = self.tensor_table(net)
        ends = [off for _, _, off in tt][1:] + [self.param_count(net)]
        return [(off, end - off) for (_, _, off), end in zip(tt, ends)][::-1]

    def all_reduce_grads(self, net, group=None):
        from rsrgan_amd import dist as rdist
        if self.bucketed:
            self.waited = getattr(self, "waited", [])
            rdist.all_reduce_mean_buckets_(self._grads[net], self.grad_buckets(net), group,
                                           wait_bucket=lambda i, stream: self.waited.append((net, i)))
        else:
            rdist.all_reduce_mean_(self._grads[net], group)

    def _np(self, a, dt=np.float64):
        return None if a is None else np.asarray(a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a, dt)

    def _store(self, net, grads):
        self._grads[net].copy_(torch.from_numpy(np.concatenate([grads[n].reshape(-1) for n, _ in self.specs[net]])))

    def forward_g(self, x, lengths):
        return torch.from_numpy(self.o.forward(self._np(x), self._np(lengths, np.int32)).astype(np.float32))

    def d_backward(self, x, lab, ln, noise_real=None, noise_fake=None, train=True, apply=False):
        losses, grads = self.o.d_tower(self._np(x), self._np(lab), self._np(ln, np.int32), self._np(noise_real),
                                       self._np(noise_fake), want_grads=train)
        if train:
            self._store(NET_D, grads)
            if apply:
                self.apply(NET_D)
        return torch.tensor(losses, dtype=torch.float32)

    def g_backward(self, x, lab, ln, noise_fake=None, train=True, reuse=False, apply=False):
        losses, grads, _ = self.o.g_tower(self._np(x), self._np(lab), self._np(ln, np.int32), self._np(noise_fake),
                                          want_grads=train)
        if train:
            self._store(NET_G, grads)
            if apply:
                self.apply(NET_G)
        return torch.tensor(losses, dtype=torch.float32)

    def apply(self, net):
        g = split_flat(self._grads[net].numpy(), self.tensor_table(net))
        (self.o.apply_g if net == NET_G else self.o.apply_d)(g)


def build_hip_pair(cfg, B, Tmax, seed=0, flags=0, **argkw):
    """(GAN_RNN on the HIP engine, fp64 oracle) with identical fp32-rounded weights."""
    from rsrgan_amd import GAN_RNN
    g, d = rand_params(cfg, seed)
    args = args_for(cfg, B, **argkw)
    model = GAN_RNN(None, args, ["gpu:0"], max_frames=Tmax, net_overrides=dict(overrides(cfg), flags=flags))
    want_g = [(n, tuple(s)) for n, s in O.g_param_specs(cfg)]
    want_d = [(n, tuple(s)) for n, s in O.d_param_specs(cfg)]
    assert [(n, s) for n, s, _ in model.engine.tensor_table(NET_G)] == want_g
    assert [(n, s) for n, s, _ in model.engine.tensor_table(NET_D)] == want_d
    model.set_vars(g, d)
    keep = float(getattr(args, "keep_prob", 1.0))
    drop = {} if keep >= 1.0 else dict(        # the oracle is fed the masks the device draws (GAN_RNN's default seed, rank 0)
        keep_prob=float(np.float32(keep)),
        mask_fn=lambda run, tower, layer, b, t, p: seq_dropout_mask(4321, run, layer, b, t, p, keep))
    oracle = O.GanRnnOracle(cfg, g, d, batch_size=B, l2_scale=args.l2_scale,
                            g_learning_rate=float(np.float32(args.g_learning_rate)),
                            d_learning_rate=float(np.float32(args.d_learning_rate)),
                            mse_lambda=float(np.float32(args.init_mse_weight)), **drop)
    return model, oracle


# ---- dropout masks of the HIP path (csrc/kernels.hip k_dropout_fwd, csrc/dnn.cpp Model::drop_key): a counter-based hash, restated
# here so that the oracle (which takes masks as an input) and the device draw the same ones ----
_M64 = (1 << 64) - 1


def _splitmix64_int(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def dropout_mask(seed, run, net, layer, call, rows, cols, keep_prob):
    return _mask_of_tag(seed, run, (net << 16) | (layer << 8) | call, rows, cols, keep_prob)


def seq_dropout_mask(seed, run, layer, B, T, P, keep_prob):
    """DropoutWrapper masks of the sequence generators (csrc/kernels.h DropSpec; csrc/model.cpp g_chain: tag = 1<<40 | layer<<20 | t,
    element = row * P + col) as a [B, T, P] array"""
    return np.stack([_mask_of_tag(seed, run, (1 << 40) | (layer << 20) | t, B, P, keep_prob) for t in range(T)], axis=1)


def _mask_of_tag(seed, run, tag, rows, cols, keep_prob):
    key = _splitmix64_int((_splitmix64_int((seed ^ ((run * 0xD1342543DE82EF95) & _M64)) & _M64) + tag) & _M64)
    with np.errstate(over="ignore"):
        x = np.uint64(key) + np.arange(rows * cols, dtype=np.uint64)
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    thr = int(float(np.float32(keep_prob)) * 16777216.0)
    return ((x >> np.uint64(40)) < np.uint64(thr)).reshape(rows, cols).astype(np.float64)


def bench_engine(batch_size, max_frames, rank):
    """RSRGAN_BENCH_ENGINE=tests.helpers:bench_engine -- the CPU stand-in bench.py's `--backend gloo` plumbing test runs on
    (tests/test_bench_ranks.py).  Identical variables on every rank, like the replicas of the product."""
    cfg = small_cfg()
    g, d = rand_params(cfg, 5)
    return OracleEngine(cfg, g, d, batch_size)
