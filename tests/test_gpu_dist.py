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
    finally:
        dist.destroy_process_group()
