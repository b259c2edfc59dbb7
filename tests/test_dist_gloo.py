"""N>1 path on CPU: world_size-2 gloo.  Rank k takes batch rows [B*k, B*(k+1))
(models/gan_rnn_placeholder.py:157-159), gradients are averaged with all-reduce
(utils/ops.py:343-376), then clipped and applied -- must equal the oracle run as 2 in-graph towers."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import rsrgan_oracle as O
from tests.helpers import OracleEngine, args_for, rand_batch, rand_params, small_cfg


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out, bucketed=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rsrgan_amd import GAN_RNN, train_one_iteration
        cfg = small_cfg()
        B, T = 2, 5
        g, d = rand_params(cfg, 5)
        lr_g, lr_d = 8e-5 * world, 1e-3 * world                    # LR x num_gpu (train...py:458-459)
        m = GAN_RNN(None, args_for(cfg, B, num_gpu=world, g_learning_rate=lr_g, d_learning_rate=lr_d, gen_updates=2),
                    ["cpu:%d" % rank], engine=OracleEngine(cfg, g, d, B))
        m.engine.bucketed = bucketed
        batches = [rand_batch(cfg, B * world, T, 60 + i, ragged=True) for i in range(2)]
        res = train_one_iteration(None, m, len(batches) * world, 0, [[None] + list(b) for b in batches])
        flat = np.concatenate([m.engine.o.g[n].reshape(-1) for n, _ in O.g_param_specs(cfg)] +
                              [m.engine.o.d[n].reshape(-1) for n, _ in O.d_param_specs(cfg)])
        out[rank] = (res, flat)
        if bucketed:          # every bucket of every averaged backward was waited for, in completion order
            nb = len(m.engine.grad_buckets(0))
            assert m.engine.waited[:nb] != [] and [i for n, i in m.engine.waited if n == 0][:nb] == list(range(nb))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucketed", [False, True])
def test_two_rank_gloo_equals_two_tower_oracle(bucketed):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out, bucketed), nprocs=world, join=True)
    cfg = small_cfg()
    B, T = 2, 5
    g, d = rand_params(cfg, 5)
    ref = O.GanRnnOracle(cfg, g, d, batch_size=B, num_towers=world,
                         g_learning_rate=float(np.float32(8e-5 * world)), d_learning_rate=float(np.float32(1e-3 * world)))
    batches = [rand_batch(cfg, B * world, T, 60 + i, ragged=True) for i in range(2)]
    want = O.train_one_iteration(ref, batches, 1, 2)
    want_flat = np.concatenate([ref.g[n].reshape(-1) for n, _ in O.g_param_specs(cfg)] +
                               [ref.d[n].reshape(-1) for n, _ in O.d_param_specs(cfg)])
    for r in range(world):
        res, flat = out[r]
        assert np.allclose(res, want, rtol=1e-6), (r, res, want)
        assert np.allclose(flat, want_flat, rtol=1e-9, atol=1e-12), r
    assert np.array_equal(out[0][1], out[1][1])                     # replicas stay bit-identical


def _rank0_worker(rank, world, port, out, fail):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rsrgan_amd import dist as rdist

        def job():
            if fail:
                raise OSError("disk full")
            return "written"
        try:
            out[rank] = ("ok", rdist.run_on_rank0(job))
        except Exception as e:                                      # noqa: BLE001
            out[rank] = ("raised", "%s: %s" % (type(e).__name__, e))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fail", [False, True])
def test_rank0_section_releases_the_other_ranks(fail):
    """Checkpoint writes and decode run on rank 0 only (Model.save, run_gan_*.main).  The waiting ranks used to sit in a barrier that a
    failing rank 0 never reached: run_on_rank0 broadcasts rank 0's status on every path and every rank fails the same way."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rank0_worker, args=(world, _free_port(), out, fail), nprocs=world, join=True)
    if fail:
        assert out[0] == ("raised", "OSError: disk full")
        assert out[1][0] == "raised" and "rank 0 failed: OSError: disk full" in out[1][1]
    else:
        assert out[0] == ("ok", "written") and out[1] == ("ok", None)
