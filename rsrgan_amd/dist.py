"""Data parallelism for the GAN step: one process per GPU, torch.distributed (backend "nccl" is RCCL
on ROCm, over xGMI; "gloo" on CPU for tests).

The reference's only parallelism is in-graph tower replication: the fed batch is sliced per GPU
(models/gan_rnn_placeholder.py:157-159), every variable's gradient is averaged over towers
(utils/ops.py:343-376), THEN clipped per tensor and applied once (:177-184), with the learning
rates multiplied by num_gpu (scripts/train_gan_rnn_placeholder.py:458-459).  Here the mean is one
all-reduce of the flat, zero-padded gradient buffer: 23.4 MB (G) / 0.75 MB (D) fp32 per step.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def is_dist(group=None) -> bool:
    return dist.is_available() and dist.is_initialized()


def world_size(group=None) -> int:
    return dist.get_world_size(group) if is_dist(group) else 1


def rank(group=None) -> int:
    return dist.get_rank(group) if is_dist(group) else 0


def init_from_env(backend: Optional[str] = None):
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws <= 1:
        return 0, 0, 1
    rk, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", os.environ["RANK"]))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rk, world_size=ws)
    return rk, local, ws


def all_reduce_mean_(flat: torch.Tensor, group=None) -> torch.Tensor:
    """average_gradients (utils/ops.py:343-376) over ranks, in place."""
    ws = world_size(group)
    if ws > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.mul_(1.0 / ws)
    return flat


def all_reduce_mean_buckets_(flat: torch.Tensor, buckets, group=None, wait_bucket=None, comm_stream=None, timing=None) -> torch.Tensor:
    """average_gradients bucket by bucket: `buckets` = (offset, count) ranges of `flat` in the order the backward pass
    completes them.  `wait_bucket(i, stream)` makes `stream` wait until range i is final (HipEngine: rsrgan_grad_bucket_wait);
    with a `comm_stream` every all-reduce is issued there, so bucket i travels while the later buckets are still being
    computed on the caller's stream, which joins the communication stream at the end."""
    if comm_stream is None:
        for i, (off, cnt) in enumerate(buckets):
            if wait_bucket is not None:
                wait_bucket(i, None)
            all_reduce_mean_(flat[off:off + cnt], group)
        return flat
    cur = torch.cuda.current_stream(flat.device)
    for i, (off, cnt) in enumerate(buckets):
        if wait_bucket is not None:
            wait_bucket(i, comm_stream)
        with torch.cuda.stream(comm_stream):
            if timing is not None:          # diagnostic (bench.py --gpus N): when each bucket's all-reduce ran on the communication stream
                e0 = torch.cuda.Event(enable_timing=True); e0.record(comm_stream)
            all_reduce_mean_(flat[off:off + cnt], group)
            if timing is not None:
                e1 = torch.cuda.Event(enable_timing=True); e1.record(comm_stream)
                timing["buckets"].append((i, 4 * cnt, e0, e1))
    if timing is not None:                  # the compute stream is done with every gradient here; what it waits for next is exposed
        timing["ready"] = torch.cuda.Event(enable_timing=True); timing["ready"].record(cur)
    cur.wait_stream(comm_stream)
    if timing is not None:
        timing["joined"] = torch.cuda.Event(enable_timing=True); timing["joined"].record(cur)
    return flat


def all_gather_rows(v: torch.Tensor, group=None) -> torch.Tensor:
    """[k] -> [world, k]: the per-tower loss lists the reference fetches (:262-268)."""
    ws = world_size(group)
    if ws == 1:
        return v.unsqueeze(0)
    out = [torch.empty_like(v) for _ in range(ws)]
    dist.all_gather(out, v.contiguous(), group=group)
    return torch.stack(out, 0)


def barrier(group=None):
    if is_dist(group):
        dist.barrier(group=group)


def run_on_rank0(fn, group=None):
    """fn() on rank 0 only (one writer: checkpoints, decode output) while the other ranks wait.  The wait is a broadcast of rank 0's
    status, reached on every path: if fn raises, rank 0 re-raises AFTER telling the others, and they raise too instead of sitting in
    a barrier until the process-group timeout.  Returns fn's result on rank 0, None elsewhere."""
    out, err = None, None
    if rank(group) == 0:
        try:
            out = fn()
        except BaseException as e:          # noqa: BLE001 -- re-raised below, after the other ranks have been released
            err = e
    if is_dist(group):
        status = [None if err is None else "%s: %s" % (type(err).__name__, err)]
        dist.broadcast_object_list(status, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if err is None and status[0] is not None:
            raise RuntimeError("rank 0 failed: " + status[0])
    if err is not None:
        raise err
    return out
