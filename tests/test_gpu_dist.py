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


def _two_rank_worker(rank, world, port, out, ref_size=False):
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
        cfg, B, T = _two_rank_case(ref_size)
        g, d = rand_params(cfg, 5)
        m = GAN_RNN(None, args_for(cfg, B, num_gpu=world, g_learning_rate=8e-5 * world, d_learning_rate=1e-3 * world, gen_updates=2),
                    ["gpu:0"], max_frames=T, net_overrides=dict(overrides(cfg), flags=3))
        m.set_vars(g, d)
        assert len(m.engine.grad_buckets(NET_G)) > 1                 # the bucketed path is the one under test
        batches = [rand_batch(cfg, B * world, T, 60 + i, ragged=True) for i in range(3)]
        if ref_size:
            m.engine.profile_launches()
        res = train_one_iteration(None, m, len(batches) * world, 0, [[None] + list(b) for b in batches])
        status = m.engine.device_status()
        gv, dv = m.get_vars()
        out[rank] = (res, {k: v.copy() for k, v in gv.items()}, {k: v.copy() for k, v in dv.items()}, status)
    finally:
        dist.destroy_process_group()


def _two_rank_case(ref_size):
    if not ref_size:
        return small_cfg("lstm"), 3, 7
    from oracle import rsrgan_oracle as O
    return O.NetCfg(), 32, 9            # the reference's networks, one 32-row group per rank: every recurrence a persistent launch


@pytest.mark.parametrize("ref_size", [False, True])
def test_two_ranks_on_one_gpu_equal_two_towers(ref_size):
    """SURVEY 8e determinism check on the HIP engine with world_size 2: must equal the oracle run as 2 in-graph towers and the
    1-rank HIP run on the concatenated batch (tower mean of tower-mean gradients = gradient of the overall mean), replicas
    bit-identical (models/gan_rnn_placeholder.py:157-184).  ref_size: the reference's networks at B = 32 per rank -- the size at
    which the generator's recurrences are the persistent launches of csrc/gpersist.hip, so the multi-rank sequence g_backward(apply
    =False) -> bucket events -> communication stream -> join -> apply runs around k_glstm_bwd and the bucket order behind it; the two
    processes' launches (2 x 114 + the discriminator's workgroups) share one device and no bounded wait may expire."""
    import torch.multiprocessing as mp
    from oracle import rsrgan_oracle as O
    from rsrgan_amd import GAN_RNN, train_one_iteration
    from tests.helpers import args_for, overrides, rand_params
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_two_rank_worker, args=(world, port, out, ref_size), nprocs=world, join=True)
    cfg, B, T = _two_rank_case(ref_size)
    g, d = rand_params(cfg, 5)
    batches = [rand_batch(cfg, B * world, T, 60 + i, ragged=True) for i in range(3)]
    ref = O.GanRnnOracle(cfg, g, d, batch_size=B, num_towers=world,
                         g_learning_rate=float(np.float32(8e-5 * world)), d_learning_rate=float(np.float32(1e-3 * world)))
    want = O.train_one_iteration(ref, batches, 1, 2)
    for r in range(world):
        res, gv, dv, status = out[r]
        assert status == 0, (r, status)
        assert np.allclose(res, want, rtol=1e-3 if ref_size else 1e-4), (r, res, want)
        for k in ref.g:
            # (Adam moves an element by ~lr whatever its gradient's size: at the reference's sizes a few of the 5.8 M elements have
            #  gradients below fp32 noise and step the other way -- bounded by the steps taken, rare in the mean)
            if ref_size:
                assert np.abs(gv[k] - ref.g[k]).max() <= 6 * 2 * 8e-5 * world and np.abs(gv[k] - ref.g[k]).mean() <= 2e-6, k
                continue
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
    assert np.allclose(res1, out[0][0], rtol=1e-3 if ref_size else 1e-4)
    assert one.engine.device_status() == 0
    gv1, dv1 = one.get_vars()
    for k in gv1:
        if ref_size:
            assert np.abs(gv1[k] - out[0][1][k]).max() <= 6 * 2 * 8e-5 * world and np.abs(gv1[k] - out[0][1][k]).mean() <= 2e-6, k
            continue
        assert np.abs(gv1[k] - out[0][1][k]).max() <= 2e-6 * max(1.0, np.abs(gv1[k]).max()), k


@pytest.mark.parametrize("B,T", [(32, 9), (64, 100)])
def test_dp_sequence_around_the_persistent_launches(B, T):
    """VERDICT r4 item 4.  At the reference's sizes the generator's BPTT is ONE persistent launch (csrc/gpersist.hip k_glstm_bwd) and
    the gradient buckets complete behind it in another order than on the launch-per-phase path.  The multi-rank sequence
    g_backward(apply=False) -> per-bucket events -> all-reduce on the communication stream -> join -> apply must give the bits of
    the fused single-rank g_step for three updates -- with NOTHING else on the device, and with collectives of a 24 MB buffer looping on
    a second stream for the whole run (world 1: RCCL's out-of-place all-reduce is a device copy kernel; the reductions of a real
    multi-rank run cannot be had on one GPU: RCCL refuses two ranks per device).  No bounded wait may expire either way."""
    import threading
    import torch.distributed as dist
    from oracle import rsrgan_oracle as O
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1)
    stop = threading.Event()
    try:
        cfg = O.NetCfg()
        fused, _ = build_hip_pair(cfg, B, T, seed=81, flags=3)
        dp, _ = build_hip_pair(cfg, B, T, seed=81, flags=3)
        busy, _ = build_hip_pair(cfg, B, T, seed=81, flags=3)
        assert len(dp.engine.grad_buckets(NET_G)) == cfg.g_layers + 2

        def run(m, mode):
            m.engine.profile_launches()
            for it in range(3):
                x, lab, ln = rand_batch(cfg, B, T, seed=90 + it, ragged=True)
                if mode == "fused":
                    m.d_step(x, lab, ln); m.g_step(x, lab, ln, reuse_g_forward=True)
                else:
                    m.engine.d_backward(x, lab, ln, train=True, apply=False)
                    m.engine.all_reduce_grads(NET_D, force=True); m.engine.apply(NET_D)
                    m.engine.g_backward(x, lab, ln, train=True, reuse=True, apply=False)
                    m.engine.all_reduce_grads(NET_G, force=True); m.engine.apply(NET_G)
            assert m.engine.device_status() == 0
            return m.get_vars()

        va = run(fused, "fused")
        vb = run(dp, "dp")

        def comm_load():
            st = torch.cuda.Stream()
            src = torch.ones(6 << 20, device="cuda"); dst = torch.empty_like(src)
            with torch.cuda.stream(st):
                while not stop.is_set():
                    for _ in range(16):
                        dist.all_reduce(src); dist.all_gather_into_tensor(dst, src); dst.add_(src)
                    st.synchronize()
        th = threading.Thread(target=comm_load, daemon=True); th.start()
        import time; time.sleep(0.2)
        vc = run(busy, "dp")
        stop.set(); th.join(timeout=20)
        # the fused step hands the three layers' kernel gradients to ONE batched stream-K launch where the product is large enough
        # (csrc/gemm.hip launch_gemm_batch: T = 100), the bucketed sequence needs them layer by layer: same products cut at other k
        # positions, so fused and bucketed agree to fp32 rounding there (Adam then moves an element by ~lr whatever its gradient's
        # size) and bit for bit where the batch is not taken (T = 9); a busy communication stream must never change a bit
        batched = T * B >= 2048
        for p_, q_, r_ in zip(va, vb, vc):
            for k in p_:
                if batched:
                    assert np.abs(p_[k] - q_[k]).max() <= 6 * 2 * 8e-5 and np.abs(p_[k] - q_[k]).mean() <= 2e-6, ("dp", k)
                else:
                    assert np.array_equal(p_[k], q_[k]), ("dp", k)
                assert np.array_equal(q_[k], r_[k]), ("dp, busy communication stream", k)
        # the persistent launches are what ran: one k_glstm_fwd + one k_glstm_bwd per iteration
        dp.engine.profile_begin()
        x, lab, ln = rand_batch(cfg, B, T, seed=99, ragged=True)
        dp.engine.d_backward(x, lab, ln, train=True, apply=False)
        dp.engine.g_backward(x, lab, ln, train=True, reuse=True, apply=False)
        n = dp.engine.profile_read_kind(1)[0] + dp.engine.profile_read_kind(3)[0] + dp.engine.profile_read_kind(2)[0]      # (kind 3: the D-run's forward under RSRGAN_DPIPE=1)
        dp.engine.profile_read()
        assert n == 2, n
    finally:
        stop.set()
        dist.destroy_process_group()
