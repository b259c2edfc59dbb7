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


def _summary_worker(rank, world, port, out, save_dir):
    """train_one_iteration + eval_one_iteration with a TensorBoard writer on rank 0 (save_dir set): the summary fetch runs on rank 0
    alone, so it must not contain a collective -- the other rank is already in the next step's gradient all-reduce."""
    import datetime
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        from rsrgan_amd import GAN_RNN, eval_one_iteration, train_one_iteration
        cfg = small_cfg()
        B, T = 2, 5
        g, d = rand_params(cfg, 5)
        m = GAN_RNN(None, args_for(cfg, B, num_gpu=world, save_dir=save_dir), ["cpu:%d" % rank], engine=OracleEngine(cfg, g, d, B))
        assert (m.writer is not None) == (rank == 0)
        batches = [rand_batch(cfg, B * world, T, 60 + i, ragged=True) for i in range(3)]
        q = [[None] + list(b) for b in batches]
        res = train_one_iteration(None, m, len(batches) * world, 0, q)          # batch 0 is summarised (batch % 100 == 0)
        ev = eval_one_iteration(None, m, len(batches) * world, 0, q)            # the last fed (global) batch is summarised
        m.save(save_dir, 1)
        flat = np.concatenate([m.engine.o.g[n].reshape(-1) for n, _ in O.g_param_specs(cfg)])
        out[rank] = (res, ev, flat)
    finally:
        dist.destroy_process_group()


def test_summaries_do_not_break_multi_rank_training(tmp_path):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_summary_worker, args=(world, _free_port(), out, str(tmp_path)), nprocs=world, join=True)
    assert np.allclose(out[0][0], out[1][0]) and np.allclose(out[0][1], out[1][1])
    assert np.array_equal(out[0][2], out[1][2])                     # the replicas stayed in step
    ev = [f for sub in ("train", "eval") for f in os.listdir(os.path.join(str(tmp_path), sub))]
    assert len(ev) == 2 and all(f.startswith("events.out.tfevents") for f in ev), ev


def _world8_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        from rsrgan_amd import GAN_RNN, train_one_iteration
        cfg = small_cfg()
        B, T = 1, 4
        g, d = rand_params(cfg, 5)
        m = GAN_RNN(None, args_for(cfg, B, num_gpu=world, g_learning_rate=8e-5 * world, d_learning_rate=1e-3 * world),
                    ["cpu:%d" % rank], engine=OracleEngine(cfg, g, d, B))
        m.engine.bucketed = True
        batches = [rand_batch(cfg, B * world, T, 80 + i, ragged=True) for i in range(2)]
        res = train_one_iteration(None, m, len(batches) * world, 0, [[None] + list(b) for b in batches])
        flat = np.concatenate([m.engine.o.g[n].reshape(-1) for n, _ in O.g_param_specs(cfg)] +
                              [m.engine.o.d[n].reshape(-1) for n, _ in O.d_param_specs(cfg)])
        out[rank] = (res, flat)
    finally:
        dist.destroy_process_group()


def test_eight_rank_gloo_equals_eight_tower_oracle():
    """The 8-GPU half of BASELINE.json configs[2] on CPU: 8 ranks, the bucketed all-reduce path, ragged shards (one utterance per rank),
    lr x 8 (train...py:458-459) == the oracle run as 8 in-graph towers (utils/ops.py:343-376, gan_rnn_placeholder.py:157-184)."""
    world = 8
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_world8_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    cfg = small_cfg()
    B, T = 1, 4
    g, d = rand_params(cfg, 5)
    ref = O.GanRnnOracle(cfg, g, d, batch_size=B, num_towers=world,
                         g_learning_rate=float(np.float32(8e-5 * world)), d_learning_rate=float(np.float32(1e-3 * world)))
    batches = [rand_batch(cfg, B * world, T, 80 + i, ragged=True) for i in range(2)]
    want = O.train_one_iteration(ref, batches, 1, 1)
    want_flat = np.concatenate([ref.g[n].reshape(-1) for n, _ in O.g_param_specs(cfg)] +
                               [ref.d[n].reshape(-1) for n, _ in O.d_param_specs(cfg)])
    for r in range(world):
        res, flat = out[r]
        assert np.allclose(res, want, rtol=1e-6), (r, res, want)
        assert np.allclose(flat, want_flat, rtol=1e-9, atol=1e-12), r
        assert np.array_equal(flat, out[0][1])                      # replicas stay bit-identical


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


class _StatEngine:
    """a stand-in with the three engine calls GAN.sync_batch_norm_state uses"""

    def __init__(self, rank):
        import torch
        names = ["g_model/fully_connected/weights", "g_model/fully_connected/BatchNorm/beta", "g_model/fully_connected/BatchNorm/moving_mean",
                 "g_model/fully_connected/BatchNorm/renorm_stddev_weight", "g_model/fully_connected_1/weights"]
        shapes = [(3, 4), (4,), (4,), (1,), (4, 2)]
        self.table, off = [], 0
        for n, sh in zip(names, shapes):
            self.table.append((n, sh, off)); off += int(np.prod(sh))
        self.flat = {0: torch.arange(off, dtype=torch.float32) + 100.0 * rank, 1: torch.zeros(0)}
        self.sets = 0

    def tensor_table(self, net):
        return self.table if net == 0 else []

    def get_params(self, net, what="variables"):
        return self.flat[net].clone()

    def set_params(self, net, flat, what="variables"):
        self.flat[net] = flat.clone(); self.sets += 1


def _bn_sync_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rsrgan_amd.gan import GAN
        m = GAN.__new__(GAN)
        m.batch_norm, m.process_group, m.engine = True, None, _StatEngine(rank)
        n = m.sync_batch_norm_state()
        out[rank] = (n, m.engine.flat[0].numpy().copy(), m.engine.sets)
    finally:
        dist.destroy_process_group()


def test_batch_norm_statistics_are_averaged_over_ranks():
    """rsrgan_amd/gan.py sync_batch_norm_state: the moving / renorm statistics (not beta, gamma or the weights) become the mean over
    the ranks on every rank -- the reference's towers update one shared copy (models/gan.py:139-146)."""
    import torch.multiprocessing as mp
    port, world = _free_port(), 2
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_bn_sync_worker, args=(world, port, out), nprocs=world, join=True)
    base = np.arange(29, dtype=np.float32)
    for rank in range(world):
        n, flat, sets = out[rank]
        assert n == 5 and sets == 1
        want = base + 100.0 * rank
        want[16:21] = base[16:21] + 50.0                    # moving_mean (4) + renorm_stddev_weight (1): the mean of +0 and +100
        assert np.array_equal(flat, want), (rank, flat, want)
