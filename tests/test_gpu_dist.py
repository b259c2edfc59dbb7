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
    finally:
        dist.destroy_process_group()
