"""Experiment: do two independent recurrences on two HIP streams overlap on MI355X?  One GAN_RNN at B=64 vs one at B=32 vs two
B=32 models stepping concurrently on their own streams (the step is replayed from hipGraphs, so the host is not the limit).
Run on the MI355X box from the repo root: python tools/two_stream.py [T]"""
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, ".")
from rsrgan_amd import GAN_RNN


def make(B, T, seed):
    args = SimpleNamespace(batch_size=B, input_dim=257, output_dim=40, left_context=0, right_context=0, g_type="lstm",
                           keep_prob=1.0, batch_norm=False, num_gpu=1, save_dir=None, l2_scale=0.0, disc_updates=1, gen_updates=1,
                           init_mse_weight=10.0, init_disc_noise_std=0.0, d_learning_rate=1e-3, g_learning_rate=8e-5)
    m = GAN_RNN(None, args, ["gpu:0"], max_frames=T, seed=seed, net_overrides=dict(flags=3))
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(rng.standard_normal((B, T, 257)).astype(np.float32)).to(dev)
    lab = torch.from_numpy(rng.standard_normal((B, T, 40)).astype(np.float32)).to(dev)
    ln = torch.from_numpy(np.full(B, T, np.int32)).to(dev)
    return m, (x, lab, ln)


def step(m, b):
    with torch.cuda.stream(m.engine.stream):
        m.d_step(*b, sync=False, gather=False)
        m.g_step(*b, reuse_g_forward=True, sync=False, gather=False)


def bench(models, steps=30, warmup=6):
    for _ in range(warmup):
        for m, b in models:
            step(m, b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for m, b in models:
            step(m, b)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps, host * 1e3 / steps


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    a64 = make(64, T, 1)
    t, h = bench([a64])
    print("one model  B=64: %.3f ms/step (host issue %.3f) -> %.0f frames/s" % (t, h, 64 * T / t * 1e3))
    del a64
    a = make(32, T, 2); b = make(32, T, 3)
    t, h = bench([a])
    print("one model  B=32: %.3f ms/step (host issue %.3f) -> %.0f frames/s" % (t, h, 32 * T / t * 1e3))
    t, h = bench([a, b])
    print("two models B=32 on two streams: %.3f ms per pair of steps (host issue %.3f) -> %.0f frames/s" % (t, h, 64 * T / t * 1e3))
    c = make(32, T, 4); d = make(32, T, 5)
    t, h = bench([a, b, c, d])
    print("four models B=32 on four streams: %.3f ms per 4 steps (host issue %.3f) -> %.0f frames/s" % (t, h, 128 * T / t * 1e3))


if __name__ == "__main__":
    main()
