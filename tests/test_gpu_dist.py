"""Data-parallel plumbing on a real GPU (1 rank of RCCL): the library's gradient buffer is aliased
zero-copy by torch, all-reduced with backend nccl (= RCCL) and applied; must equal the fused step."""
import os
import socket

import numpy as np
import pytest
import torch

from tests.helpers import NET_D, NET_G, build_hip_pair, rand_batch, small_cfg

pytestmark = pytest.mark.gpu


def test_grad_view_aliases_library_buffer():
    cfg = small_cfg()
    model, oracle = build_hip_pair(cfg, 4, 6, seed=61)
    x, lab, ln = rand_batch(cfg, 4, 6, seed=62)
    model.engine.d_backward(x, lab, ln, train=True, apply=False)
    v = model.engine.grad_view(NET_D)
    assert v.is_cuda and v.dtype == torch.float32 and v.numel() >= model.engine.param_count(NET_D)
    dense0 = model.engine.get_grads(NET_D).clone()
    assert float(dense0.abs().sum()) > 0
    v.mul_(2.0)                                           # writes through to the library's buffer
    torch.cuda.synchronize()
    assert torch.allclose(model.engine.get_grads(NET_D), 2.0 * dense0)
    assert abs(float(v.abs().sum()) - float((2 * dense0).abs().sum())) < 1e-3 * float(v.abs().sum())   # padding is zero


def test_nccl_world1_dp_path_equals_fused_step():
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        from rsrgan_amd import dist as rdist
        cfg = small_cfg()
        a, oracle = build_hip_pair(cfg, 4, 6, seed=63)
        b, _ = build_hip_pair(cfg, 4, 6, seed=63)
        x, lab, ln = rand_batch(cfg, 4, 6, seed=64, ragged=True)
        la = a.d_step(x, lab, ln)                                         # fused: backward + apply
        lb = b.engine.d_backward(x, lab, ln, train=True, apply=False)     # DP: backward | all-reduce | apply
        rdist.all_reduce_mean_(b.engine.grad_view(NET_D))
        b.engine.apply(NET_D)
        ga = a.g_step(x, lab, ln)
        b.engine.g_backward(x, lab, ln, train=True, apply=False)
        rdist.all_reduce_mean_(b.engine.grad_view(NET_G))
        b.engine.apply(NET_G)
        va, vb = a.get_vars(), b.get_vars()
        for p, q in zip(va, vb):
            for k in p:
                assert np.array_equal(p[k], q[k]), k
        assert np.allclose(np.ravel(la), lb.cpu().numpy())
        assert rdist.all_gather_rows(lb).shape == (1, 3)
        # bucketed path (wavefront schedule): per-bucket events + communication stream, forced at world size 1
        for g_type in ("lstm", "res_lstm_l"):
            cfg2 = small_cfg(g_type)
            c, _ = build_hip_pair(cfg2, 4, 6, seed=65, flags=1)
            d, _ = build_hip_pair(cfg2, 4, 6, seed=65, flags=1)
            bk = c.engine.grad_buckets(NET_G)
            assert len(bk) == cfg2.g_layers + (2 if g_type == "lstm" else 1)
            assert sorted(bk)[0][0] == 0 and sum(n for _, n in bk) == c.engine.grad_view(NET_G).numel()
            ends = sorted((o, o + n) for o, n in bk)
            assert all(ends[i][1] == ends[i + 1][0] for i in range(len(ends) - 1))       # the ranges tile the buffer
            assert c.engine.grad_buckets(NET_D) == [(0, c.engine.grad_view(NET_D).numel())]
            for it in range(3):
                xs, ls, lns = rand_batch(cfg2, 4, 6, seed=70 + it, ragged=True)
                c.d_step(xs, ls, lns); d.d_step(xs, ls, lns)
                c.g_step(xs, ls, lns, reuse_g_forward=True)                                # fused
                d.engine.g_backward(xs, ls, lns, train=True, reuse=True, apply=False)     # DP, bucket by bucket
                d.engine.all_reduce_grads(NET_G, force=True)
                d.engine.apply(NET_G)
            for p_, q_ in zip(c.get_vars(), d.get_vars()):
                for k in p_:
                    assert np.array_equal(p_[k], q_[k]), (g_type, k)
            # the per-bucket diagnostic bench.py prints at --gpus N: one row per bucket, in completion order, nothing negative
            d.engine.bucket_timing = True
            d.engine.g_backward(xs, ls, lns, train=True, reuse=False, apply=False)
            d.engine.all_reduce_grads(NET_G, force=True)
            rep = d.engine.bucket_report()
            d.engine.apply(NET_G)
            assert [r["bucket"] for r in rep["buckets"]] == list(range(len(bk)))
            assert sum(r["bytes"] for r in rep["buckets"]) == 4 * d.engine.grad_view(NET_G).numel()
            assert rep["exposed_ms"] >= 0 and all(r["allreduce_ms"] >= 0 for r in rep["buckets"])
    finally:
        dist.destroy_process_group()


def _two_rank_worker(rank, world, port, out):
    """Two processes share cuda:0; gradients travel through gloo (RCCL refuses two ranks on one device), everything else is the
    real HIP engine: sharding, per-bucket events + communication stream, average, clip, apply."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rsrgan_amd import GAN_RNN, train_one_iteration
        from tests.helpers import args_for, overrides, rand_params
        cfg = small_cfg("lstm")
        B, T = 3, 7
        g, d = rand_params(cfg, 5)
        m = GAN_RNN(None, args_for(cfg, B, num_gpu=world, g_learning_rate=8e-5 * world, d_learning_rate=1e-3 * world, gen_updates=2),
                    ["gpu:0"], max_frames=T, net_overrides=dict(overrides(cfg), flags=3))
        m.set_vars(g, d)
        assert len(m.engine.grad_buckets(NET_G)) > 1                 # the bucketed path is the one under test
        batches = [rand_batch(cfg, B * world, T, 60 + i, ragged=True) for i in range(3)]
        res = train_one_iteration(None, m, len(batches) * world, 0, [[None] + list(b) for b in batches])
        gv, dv = m.get_vars()
        out[rank] = (res, {k: v.copy() for k, v in gv.items()}, {k: v.copy() for k, v in dv.items()})
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_two_towers():
    """SURVEY 8e determinism check on the HIP engine with world_size 2: must equal the oracle run as 2 in-graph towers and the
    1-rank HIP run on the concatenated batch (tower mean of tower-mean gradients = gradient of the overall mean), replicas
    bit-identical (models/gan_rnn_placeholder.py:157-184)."""
    import torch.multiprocessing as mp
    from oracle import rsrgan_oracle as O
    from rsrgan_amd import GAN_RNN, train_one_iteration
    from tests.helpers import args_for, overrides, rand_params
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_two_rank_worker, args=(world, port, out), nprocs=world, join=True)
    cfg = small_cfg("lstm")
    B, T = 3, 7
    g, d = rand_params(cfg, 5)
    batches = [rand_batch(cfg, B * world, T, 60 + i, ragged=True) for i in range(3)]
    ref = O.GanRnnOracle(cfg, g, d, batch_size=B, num_towers=world,
                         g_learning_rate=float(np.float32(8e-5 * world)), d_learning_rate=float(np.float32(1e-3 * world)))
    want = O.train_one_iteration(ref, batches, 1, 2)
    for r in range(world):
        res, gv, dv = out[r]
        assert np.allclose(res, want, rtol=1e-4), (r, res, want)
        for k in ref.g:
            assert np.abs(gv[k] - ref.g[k]).max() <= 1e-5 * max(1.0, np.abs(ref.g[k]).max()), k
        for k in ref.d:
            assert np.abs(dv[k] - ref.d[k]).max() <= 1e-5 * max(1.0, np.abs(ref.d[k]).max()), k
    for k in out[0][1]:
        assert np.array_equal(out[0][1][k], out[1][1][k]), k                   # replicas stay bit-identical
    # one rank, concatenated batch, same learning rates
    one = GAN_RNN(None, args_for(cfg, B * world, num_gpu=1, g_learning_rate=8e-5 * world, d_learning_rate=1e-3 * world, gen_updates=2),
                  ["gpu:0"], max_frames=T, net_overrides=dict(overrides(cfg), flags=3))
    one.set_vars(g, d)
    res1 = train_one_iteration(None, one, len(batches), 0, [[None] + list(b) for b in batches])
    assert np.allclose(res1, out[0][0], rtol=1e-4)
    gv1, dv1 = one.get_vars()
    for k in gv1:
        assert np.abs(gv1[k] - out[0][1][k]).max() <= 2e-6 * max(1.0, np.abs(gv1[k]).max()), k
