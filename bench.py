#!/usr/bin/env python
"""bench.py -- GAN train frames/sec (one D-update + one G-update on the same minibatch).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json north_star / SURVEY.md 8d): synthetic [B=64, T=100, 257 -> 40] per GPU
(weak scaling: the reference defines batch_size per tower, models/gan_rnn_placeholder.py:96),
reference-true networks G = FC 257->280 + 3 x LSTMP(760, proj 280) + FC 280->40
(models/lstm.py:43-45) and D = 2 x LSTMP(256, proj 40) + FC 40->1
(models/discriminator_lstm.py:26-28), seed-1234 N(0,1) inputs, xavier-uniform seed-4321 weights,
g_lr 8e-5, d_lr 1e-3, mse_lambda 10, clip 15, noise std 0 (run_gan_rnn_placeholder.sh:124-142).
A step = disc_updates=1 D-run + gen_updates=1 G-run through the C ABI (the G-run reuses the D-run's
generator forward: G does not change in between).  Inputs are resident in HBM when the timed region
starts.  One JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic GEMM FLOP per frame of one (1 D + 1 G) step = 3*F_G + 8*F_D
PEAK_FP32_MFMA_TFLOPS = 157.3       # /opt/skills/guides/MI355X_MICROARCH.md (v_mfma_f32_32x32x2_f32)


ARITH_NOTE = ("fp32 MFMA (v_mfma_f32_16x16x4 / 32x32x2), fp32 accumulate; the ring hand-offs of the persistent generator launches "
              "(the 38-way partial sums of every step and layer: projection / state-gradient / input-gradient partials) are rounded to "
              "a 22-bit mantissa -- the lowest bit carries the ring pass's parity (RSRGAN_GP_TAGS=1, csrc/gpersist.hip gp_store_t); "
              "everything else is IEEE fp32 like the reference (gan_rnn_placeholder.py:94-104)")


def parity_margin(g_type, B, T):
    """ACHIEVED error of the HIP path against the fp64 oracle for this workload as tests/test_gpu_fullsize.py
    ::test_full_size_step_as_benched_against_oracle measured it on an MI355X (max relative error of the losses, worst gradient tensor's
    relative L2 error, enhanced-MFCC L1), for RSRGAN_DPIPE on / off and tagged / untagged hand-offs; committed by
    tools/mk_parity_margin.py.  The bounds are the north_star's 1e-3 (losses, MFCC) and the tests' 2e-3 (gradients)."""
    path = os.path.join(ROOT, "profiles", "r6_parity_margin.json")
    if not os.path.exists(path):
        return None
    try:
        doc = json.load(open(path))
    except (OSError, ValueError):
        return None
    key = "%s_B%d_T%d" % (g_type, B, T)
    rows = {k[len(key) + 1:]: v for k, v in doc.get("cases", {}).items() if k.startswith(key + "_")}
    if not rows:
        return None
    cur = "dpipe%s_tags%s" % ("0" if os.environ.get("RSRGAN_DPIPE", "0") in ("", "0") else "1", "0" if os.environ.get("RSRGAN_GP_TAGS", "1") == "0" else "1")
    sel = rows.get(cur)
    return {"this_configuration": cur, "loss": sel and max(sel["loss"], sel["loss_after_updates"]), "grad": sel and max(sel["grad_d"], sel["grad_g"]),
            "mfcc": sel and sel["mfcc_l1"], "bound": {"loss": 1e-3, "grad": 2e-3, "mfcc": 1e-3}, "all_configurations": rows,
            "source": "profiles/r6_parity_margin.json (%s)" % doc.get("measured", "")}


def flop_per_frame(din, dout, g_type, gl, gh, gp, dl_, dh, dp):
    lstmp = lambda i, h, p: 2 * ((i + p) * 4 * h + h * p) if p > 0 else 2 * (i + h) * 4 * h     # p == 0: num_proj=None
    gp_out = gp if gp > 0 else gh
    fc = lambda i, o: 2 * i * o
    if g_type == "lstm":
        fg = fc(din, gp_out) + gl * lstmp(gp_out, gh, gp) + fc(gp_out, dout)
    else:
        fg = lstmp(din, gh, gp) + (gl - 1) * lstmp(gp_out, gh, gp) + fc(gp_out, dout)
    fd = lstmp(dout, dh, dp) + (dl_ - 1) * lstmp(dp, dh, dp) + fc(dp, 1)
    return 3 * fg + 8 * fd, fg, fd


def synthetic(B, T, din, dout, seed=1234):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, T, din)).astype(np.float32)
    lab = rng.standard_normal((B, T, dout)).astype(np.float32)
    ln = np.full(B, T, np.int32)
    return x, lab, ln


def _usable_cpus():
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (os.cpu_count() reports the
    host's cores even inside a quota-limited container, and one torch thread per reported core then oversubscribes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return n


def _cpu_run(net, B, T, threads, budget_s, max_steps):
    """`max_steps` (or as many as fit `budget_s`, at least one) full (1D+1G) steps of the oracle's torch-CPU twin."""
    from oracle import rsrgan_oracle as O
    from oracle import torch_twin as TT
    torch.set_num_threads(threads)
    cfg = O.NetCfg() if net == "lstm" else O.NetCfg.res_lstm_l()
    g = O.xavier_init(O.g_param_specs(cfg), np.random.default_rng(4321), np.float32)
    d = O.xavier_init(O.d_param_specs(cfg), np.random.default_rng(4322), np.float32)
    tw = TT.GanRnnTorchTwin(cfg, g, d)
    x, lab, ln = synthetic(B, T, cfg.input_dim, cfg.output_dim)
    t0 = time.time(); n = 0
    while True:
        tw.d_step(x, lab, ln); tw.g_step(x, lab, ln); n += 1
        if n >= max_steps or time.time() - t0 > budget_s:
            break
    return n, time.time() - t0


def cpu_baseline(net, B, T, budget_s=9.0):
    """The oracle's torch-CPU twin timed on this box's host cores on a bounded sample of the SAME workload (kind 'port': the
    reference's TF-1.4 path cannot run here), at min(16, cores) threads (the headline: the per-step GEMMs have M = 64 rows and
    stop scaling there), at 1 thread and at all usable cores.  The 1-thread and all-core legs run in child
    processes with a hard time limit (one thread per core on M = 64 GEMMs can take minutes per step on a many-core host)."""
    import subprocess
    ncpu = _usable_cpus()
    head = min(16, ncpu)
    n, dt = _cpu_run(net, B, T, head, budget_s, 12)

    def child(threads, limit_s):
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", "%s,%d,%d,%d" % (net, B, T, threads)]
        try:
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=limit_s, env=dict(os.environ, OMP_NUM_THREADS=str(threads)))
            r = json.loads(out.stdout.strip().splitlines()[-1])
            return {"value": round(B * T * r["steps"] / r["seconds"], 1), "cores": threads, "sample": "%d step(s)" % r["steps"]}
        except subprocess.TimeoutExpired:
            return {"value": None, "cores": threads, "sample": "one step did not finish within %d s" % limit_s}
        except Exception as e:           # never lose the bench line to a baseline leg
            return {"value": None, "cores": threads, "sample": "failed: %s" % str(e)[:120]}

    one = child(1, 75)
    allc = ({"value": round(B * T * n / dt, 1), "cores": ncpu, "sample": "%d steps" % n} if ncpu == head else child(ncpu, 45))
    model_name = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model_name = line.split(":", 1)[1].strip(); break
    except OSError:
        pass
    return {"value": round(B * T * n / dt, 1), "unit": "frames/s", "cores": head, "kind": "port",
            "sample": "%d full (1D+1G) steps of the same [B=%d,T=%d] workload, oracle/torch_twin.py fp32, "
                      "os.cpu_count=%s, usable=%d, cpu=%s" % (n, B, T, os.cpu_count(), ncpu, model_name),
            "one_core": one, "all_cores": allc}


def hbm_activity(step, ms_per_step, seconds=4.0):
    """Device-level estimate of the REAL HBM traffic of the step: the memory controllers' activity (rocm-smi "GPU Memory Read/Write
    Activity (%)" = amd-smi UMC_ACTIVITY) sampled while the step loop runs un-profiled for a few seconds.  The PMC figure
    (roofline.traffic) counts the L2's fabric requests of serialised, cold-L2 dispatches, Infinity-Cache hits included: an upper
    bound.  bytes per step ~ activity x 8 TB/s (spec peak, MI355X_MICROARCH.md) x step time; the counter has 1 % resolution."""
    import re, shutil, subprocess, threading
    smi = shutil.which("rocm-smi")
    if not smi:
        return None
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            try:
                out = subprocess.run([smi, "--showmemuse"], capture_output=True, text=True, timeout=5).stdout
                m = re.search(r"Read/Write Activity \(%\):\s*([0-9.]+)", out)
                if m:
                    samples.append(float(m.group(1)))
            except Exception:
                pass
            stop.wait(0.25)
    th = threading.Thread(target=sampler, daemon=True)
    t0 = time.time()
    n = 0
    th.start()
    while time.time() - t0 < seconds:
        for _ in range(20):
            step()
        torch.cuda.synchronize(); n += 20
    stop.set(); th.join(timeout=6)
    mid = samples[1:-1] if len(samples) > 4 else samples       # the first / last sample straddle the start / end of the loop
    if not mid:
        return None
    pct = sum(mid) / len(mid)
    return {"umc_activity_pct": round(pct, 2), "samples": len(mid), "steps": n,
            "hbm_bytes_per_step_estimate": int(pct / 100.0 * 8.0e12 * ms_per_step * 1e-3),
            "how": "mean rocm-smi memory read/write activity over %d samples while %d un-profiled steps ran, x 8 TB/s x ms/step" % (len(mid), n)}


def measure_sequence(a, net, d_type, B, T, steps, warmup, rank, local, world, dev):
    """One timed run of the sequence-level GAN step: W untimed steps, then exactly `steps` steps between
    barrier + synchronize on both sides; MAX over ranks."""
    from types import SimpleNamespace
    from rsrgan_amd import GAN_RNN, dist as rdist
    g_type = "res_lstm_base" if net == "baseline_named" else net
    if net == "baseline_named":
        d_type = "dnn"
    args = SimpleNamespace(batch_size=B, input_dim=257, output_dim=40, left_context=0, right_context=0, g_type=g_type,
                           keep_prob=1.0, batch_norm=False, num_gpu=world, save_dir=None, l2_scale=0.0,
                           disc_updates=1, gen_updates=a.gen_updates, init_mse_weight=10.0, init_disc_noise_std=0.0,
                           d_learning_rate=1e-3 * world, g_learning_rate=8e-5 * world)   # LR x num_gpu (:458-459)
    ov = dict(flags=a.flags)
    if d_type == "dnn":
        ov["d_type"] = "dnn"
    if net == "baseline_named":
        ov.update(g_layers=2, g_cells=512, g_proj=0)
    model = GAN_RNN(None, args, ["gpu:%d" % local], max_frames=T, seed=4321, net_overrides=ov)
    x, lab, ln = synthetic(B, T, 257, 40, seed=1234 + rank)
    x = torch.from_numpy(x).to(dev); lab = torch.from_numpy(lab).to(dev); ln = torch.from_numpy(ln).to(dev)

    def step():
        # gather=False: like train_one_iteration, the tower mean of the loss scalars is taken once per iteration, not per step
        model.d_step(x, lab, ln, sync=False, gather=False)
        out = None
        for i in range(a.gen_updates):
            out = model.g_step(x, lab, ln, reuse_g_forward=(i == 0), sync=False, gather=False)
        return out

    with model.engine.on_stream():             # one real stream for the whole loop (hipGraph replay; no null-stream hops)
        for _ in range(warmup):
            step()
        rdist.barrier(); torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(steps):
            last = step()
            ev[i + 1].record()
        torch.cuda.synchronize(); rdist.barrier()
        dt = time.perf_counter() - t0
        dev_ms = ev[0].elapsed_time(ev[steps])
        per_step = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
        med_ms = per_step[steps // 2]
        if world > 1:
            t = torch.tensor([dt, dev_ms], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt, dev_ms = float(t[0]), float(t[1])
        losses = last.mean(0).cpu().numpy()
        if not np.all(np.isfinite(losses)):
            raise SystemExit("non-finite losses: %s" % losses)
        # one more (untimed) step with every launch of the dominant kernel bracketed by HIP events on its stream
        prof = ("k_fwd_gates", 0, 0.0, 0.0)
        if not a.no_kernel_timing:
            model.engine.profile_begin()
            step()
            prof_gp = model.engine.profile_read_kind(1)            # k_glstm_fwd: the persistent generator recurrence (csrc/gpersist.hip)
            prof_gb = model.engine.profile_read_kind(2)            # k_glstm_bwd: its BPTT
            prof = model.engine.profile_read()                    # k_fwd_gates launches of the same step
            prof = max([("k_glstm_bwd",) + tuple(prof_gb), ("k_glstm_fwd",) + tuple(prof_gp), ("k_fwd_gates",) + tuple(prof)], key=lambda p_: p_[2])
            # (the floor chain is skipped under rocprofv3: 800 extra launches would distort the committed kernel statistics)
            under_prof = any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ)
            chain = (model.engine.profile_launches(), model.engine.launch_floor(400, 0), model.engine.launch_floor(400, 1)) if not under_prof else None
        hbm = None
        if not a.no_kernel_timing and world == 1 and net == "lstm" and not a.no_hbm_activity:
            hbm = hbm_activity(step, dt * 1e3 / steps)
        buckets = None
        if world > 1:                       # one more untimed step with the bucketed gradient all-reduce timed on its communication stream
            model.engine.bucket_timing = True
            step()
            buckets = model.engine.bucket_report()
            model.engine.bucket_timing = False
    return dict(model=model, g_type=g_type, d_type=d_type, dt=dt, dev_ms=dev_ms, med_ms=med_ms, losses=losses, prof=prof,
                chain=(chain if not a.no_kernel_timing else None), buckets=buckets, hbm=hbm)


def bench_dnn_gan(a, rank, local, world, dev):
    """Frame-level GAN (models/gan.py): G = DNN 2827 -> 4x1024 -> 40, D = discriminator_dnn 297 -> 4x1024 -> 1,
    Adam/Adam.  One step = 1 D-run + 1 G-run on N = --batch frames per GPU."""
    from types import SimpleNamespace
    from rsrgan_amd import GAN, dist as rdist
    N = a.batch
    args = SimpleNamespace(batch_size=N, input_dim=257, output_dim=40, left_context=5, right_context=5, g_type="dnn",
                           keep_prob=1.0, batch_norm=bool(getattr(a, "batch_norm", False)), num_gpu=world, save_dir=None, l2_scale=0.0,
                           disc_updates=1, gen_updates=1, init_mse_weight=10.0, d_learning_rate=1e-4 * world, g_learning_rate=1e-4 * world)
    trainer = a.net == "dnn_trainer"          # BASELINE.json configs[0]: the DNN generator alone under DNNTrainer (supervised)
    if trainer:
        from rsrgan_amd.trainer import DNNTrainer
        model = DNNTrainer(None, args, ["gpu:%d" % local], seed=4321)
    else:
        model = GAN(None, args, ["gpu:%d" % local], seed=4321)
    rng = np.random.default_rng(1234 + rank)
    x = torch.from_numpy(rng.standard_normal((N, 1, 2827)).astype(np.float32)).to(dev)
    lab = torch.from_numpy(rng.standard_normal((N, 1, 40)).astype(np.float32)).to(dev)
    x2 = torch.from_numpy(rng.standard_normal((N, 1, 2827)).astype(np.float32)).to(dev)
    lab2 = torch.from_numpy(rng.standard_normal((N, 1, 40)).astype(np.float32)).to(dev)

    import contextlib
    stack = contextlib.ExitStack(); stack.enter_context(model.engine.on_stream())     # one real stream for every call below

    def step():
        if trainer:
            return model.step(x, lab, sync=False)
        # scripts/train_gan_dnn.py:50-96: the D-run and the G-run each dequeue their own batch -> no shared generator forward
        model.d_step(x, lab, sync=False)
        return model.g_step(x2, lab2, sync=False)
    for _ in range(a.warmup):
        step()
    rdist.barrier(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(a.steps):
        last = step()
    e1.record(); torch.cuda.synchronize(); rdist.barrier()
    dt = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1)
    if world > 1:
        t = torch.tensor([dt, dev_ms], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt, dev_ms = float(t[0]), float(t[1])
    if rank == 0:
        fg = 2 * (2827 * 1024 + 3 * 1024 * 1024 + 1024 * 40)
        fd = 2 * (297 * 1024 + 3 * 1024 * 1024 + 1024)
        fpf = (3 * fg) if trainer else (4 * fg + 8 * fd)   # trainer: G fwd + 2x bwd; GAN: + the D-run's own G fwd, D terms as SURVEY 8d
        ach = fpf * N / (dev_ms * 1e-3 / a.steps) / 1e12
        out = {"metric": ("supervised train frames/sec, DNN generator 2827->40 under DNNTrainer (BASELINE configs[0])" if trainer else
                          "GAN train frames/sec (G+D step), frame-level DNN-GAN 2827->40 (SURVEY 8f-1)"),
               "value": round(N * world * a.steps / dt, 1), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(dt * 1e3 / a.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": ("models/dnn_trainer.py step, G=dnn(2827-4x1024-40), " if trainer else
                                       "models/gan.py 1D+1G step, G=dnn(2827-4x1024-40)+D=discriminator_dnn(297-4x1024-1), ") +
                                      ("batch_norm(renorm=True), " if args.batch_norm else "") +
                                      "N=%d frames/GPU" % N, "global_batch": N * world, "parallelism": "dp%d" % world,
                          "losses_last_step": [round(float(v), 6) for v in last.mean(0).cpu().numpy()]},
               "roofline": {"bound": "mfma", "achieved": round(ach, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
                            "scope": "all launches of one step; algorithmic %s = %d FLOP/frame" % ("3*F_G" if trainer else "4*F_G+8*F_D", fpf)}}
        print(json.dumps(out), flush=True)
    stack.close()
    rdist.barrier()


def bench_rced(a, rank, local, world, dev):
    """R-CED under DNNTrainer (models/rced.py + dnn_trainer.py; run_dnn.sh:124-140 without batch_norm): input_dim 40 spliced
    +-5 (fed 440) -> 9 x conv2d -> FC -> 40, batch --batch frames per GPU, one Adam step per batch."""
    from types import SimpleNamespace
    from rsrgan_amd import dist as rdist
    from rsrgan_amd import GAN
    from rsrgan_amd.trainer import DNNTrainer
    N, W, S = a.batch, a.rced_width, 11
    args = SimpleNamespace(batch_size=N, input_dim=W, output_dim=40, left_context=5, right_context=5, g_type="rced", keep_prob=1.0,
                           batch_norm=bool(getattr(a, "batch_norm", False)), num_gpu=world, save_dir=None, l2_scale=0.0, g_learning_rate=1e-3 * world,
                           d_learning_rate=1e-4 * world, init_mse_weight=10.0, disc_updates=1, gen_updates=1)
    if a.rced_gan:          # BASELINE.json configs[3]: R-CED generator + discriminator_dnn (the reference's gan.py only accepts 'dnn')
        class RcedGan(GAN):
            G_TYPES = ("dnn", "rced")
        model = RcedGan(None, args, ["gpu:%d" % local], seed=4321)
        # the frame-level scripts feed the D-run and the G-run different batches (train_gan_dnn.py:50-96): no forward reuse
        model.step = lambda x_, l_, sync=False: (model.d_step(x_, l_, sync=False), model.g_step(x2, lab2, sync=False))[1]
    else:
        model = DNNTrainer(None, args, ["gpu:%d" % local], seed=4321)
    rng = np.random.default_rng(1234 + rank)
    x = torch.from_numpy(rng.standard_normal((N, 1, S * W)).astype(np.float32)).to(dev)
    lab = torch.from_numpy(rng.standard_normal((N, 1, 40)).astype(np.float32)).to(dev)
    x2 = torch.from_numpy(rng.standard_normal((N, 1, S * W)).astype(np.float32)).to(dev)
    lab2 = torch.from_numpy(rng.standard_normal((N, 1, 40)).astype(np.float32)).to(dev)
    import contextlib
    stack = contextlib.ExitStack(); stack.enter_context(model.engine.on_stream())
    for _ in range(a.warmup):
        model.step(x, lab, sync=False)
    rdist.barrier(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(a.steps):
        last = model.step(x, lab, sync=False)
    e1.record(); torch.cuda.synchronize(); rdist.barrier()
    dt = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1)
    if world > 1:
        t = torch.tensor([dt, dev_ms], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt, dev_ms = float(t[0]), float(t[1])
    if rank == 0:
        # Algorithmic FLOP per frame, two conventions.  VALID TAPS (the roofline fraction): a SAME convolution only multiplies the
        # taps that meet data -- output row h meets S - |h - (S-1)/2| of the S filter rows (91 of 121 at S = 11), output column w
        # meets the fw taps minus those beyond the frame edge (fw W - pl (pl + 1) summed over w, pl = (fw - 1) / 2) -- and the
        # kernels skip the rest (conv.hip:143-151).  INCLUDING PADDING: 2 S W (S fw Cin) Cout per layer, every tap of every position
        # (what round 3 printed as the only figure).
        num, wid, cin, fl, flp = (12, 16, 20, 24, 32, 24, 20, 16, 12), (13, 11, 9, 7, 7, 7, 9, 11, 13), 1, 0, 0
        rows_valid = sum(S - abs(h - (S - 1) // 2) for h in range(S))
        for co, fw in zip(num, wid):
            pl = (fw - 1) // 2
            fl += 2 * rows_valid * (fw * W - pl * (pl + 1)) * cin * co
            flp += 2 * S * W * (S * fw * cin) * co
            cin = co
        fl += 2 * S * W * cin * 40; flp += 2 * S * W * cin * 40
        extra = fl + 8 * 2 * ((W + 40) * 1024 + 3 * 1024 * 1024 + 1024) if a.rced_gan else 0     # the D-run's own generator forward + 8 F_D
        extra_p = extra - fl + flp if a.rced_gan else 0
        fpf = 3 * fl + extra                              # forward + data gradient + weight gradient
        fpf_p = 3 * flp + extra_p
        ach = fpf * N / (dev_ms * 1e-3 / a.steps) / 1e12
        ach_p = fpf_p * N / (dev_ms * 1e-3 / a.steps) / 1e12
        out = {"metric": ("GAN train frames/sec (G+D step), R-CED generator + discriminator_dnn (BASELINE configs[3])" if a.rced_gan
                          else "supervised train frames/sec, R-CED generator (SURVEY 8f-2)"), "value": round(N * world * a.steps / dt, 1),
               "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt * 1e3 / a.steps, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": ("models/rced.py + discriminator_dnn, 1D+1G step" if a.rced_gan else "models/rced.py under DNNTrainer") +
                                      " (batch_norm=%s), frame width %d x splice 11, N=%d frames/GPU" % (args.batch_norm, W, N),
                          "global_batch": N * world, "parallelism": "dp%d" % world,
                          "losses_last_step": [round(float(v), 6) for v in last.mean(0).cpu().numpy()]},
               "roofline": {"bound": "mfma", "achieved": round(ach, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "frac_incl_padding": round(ach_p / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
                            "scope": "all launches of one step; algorithmic %d FLOP/frame on VALID taps (%s x %d conv + FC GEMM terms%s); "
                                     "frac_incl_padding counts the zero-padding taps of the SAME convolutions too (%d FLOP/frame), which "
                                     "the kernels skip" % (fpf, "4" if a.rced_gan else "3", fl, " + 8 F_D" if a.rced_gan else "", fpf_p)}}
        print(json.dumps(out), flush=True)
    stack.close()
    rdist.barrier()


def segan_flops(L, U, depths, gk=20, dk=31):
    """algorithmic conv / dense FLOP (2 x MAC) per sample: generator forward F_G, discriminator forward F_D of ONE call"""
    le = [L]
    for _ in depths:
        le.append((le[-1] + 1) // 2)
    fg, cin = 0, 1
    for i, d in enumerate(depths):
        fg += 2 * le[i + 1] * gk * cin * d
        cin = d
    cin = 2 * depths[-1]
    dec = list(depths[:-1][::-1]) + [1]
    for j, d in enumerate(dec):                             # conv2d_transpose: every input position meets every tap once
        fg += 2 * le[len(depths) - j] * gk * cin * d
        cin = 2 * d
    fg += 2 * L * U
    ld = [L + U]
    for _ in depths:
        ld.append((ld[-1] + 1) // 2)
    fd, cin = 0, 1
    for i, d in enumerate(depths):
        fd += 2 * ld[i + 1] * dk * cin * d
        cin = d
    fd += 2 * ld[-1] * dk * cin + 2 * ld[-1]
    return fg, fd


def bench_segan(a, rank, local, world, dev):
    """BASELINE.json configs[4]: SEGAN-style conv G/D (models/segan.py) on --segan-len-sample chunks, --batch chunks per GPU; one
    step = one D-run + one G-run (scripts/train_segan.py:32-44), z and the three discriminator noise draws made on the device."""
    from types import SimpleNamespace
    from rsrgan_amd import SEGAN, dist as rdist
    from rsrgan_amd.segan import DEPTHS
    B, L, U = a.batch, a.segan_len, 40
    args = SimpleNamespace(batch_size=B, input_dim=L, output_dim=U, left_context=0, right_context=0, g_type="ae", deconv_type="deconv",
                           bias_downconv=True, bias_deconv=True, bias_D_conv=True, g_nl="prelu", init_noise_std=0.5, init_l1_weight=100.0,
                           g_learning_rate=1e-3, d_learning_rate=1e-3, save_dir=None)      # run_segan.sh:96-124
    model = SEGAN(None, args, ["gpu:%d" % local], seed=4321, process_group=(torch.distributed.group.WORLD if world > 1 else None))
    rng = np.random.default_rng(1234 + rank)
    x = torch.from_numpy(rng.standard_normal((B, L)).astype(np.float32)).to(dev)
    lab = torch.from_numpy(rng.standard_normal((B, U)).astype(np.float32)).to(dev)

    def step():
        model.d_step(x, lab)
        return model.g_step(x, lab)
    for _ in range(a.warmup):
        step()
    rdist.barrier(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(a.steps):
        last = step()
    e1.record(); torch.cuda.synchronize(); rdist.barrier()
    dt = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1)
    if world > 1:
        t = torch.tensor([dt, dev_ms], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt, dev_ms = float(t[0]), float(t[1])
    if rank == 0:
        fg, fd = segan_flops(L, U, DEPTHS)
        # D-run: G forward, three discriminator calls forward and backward (data + weight gradients); G-run: G forward, reference +
        # fake calls forward, the fake call's data gradient, G backward (data + weight gradients)
        fps = 4 * fg + 12 * fd
        ach = fps * B / (dev_ms * 1e-3 / a.steps) / 1e12
        out = {"metric": "GAN train chunks/sec (D-run + G-run), SEGAN-style conv G/D on %d-sample chunks (BASELINE configs[4])" % L,
               "value": round(B * world * a.steps / dt, 2), "unit": "chunks/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(dt * 1e3 / a.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": "models/segan.py 1 D-run + 1 G-run, AEGenerator 11 x (k=20, 16..1024) + VBN discriminator 11 x (k=31), "
                                      "%d-sample chunks -> %d, B=%d/GPU" % (L, U, B), "global_batch": B * world, "parallelism": "dp%d" % world,
                          "losses_last_step": [round(float(v), 6) for v in last]},
               "roofline": {"bound": "mfma", "achieved": round(ach, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
                            "scope": "all launches of one step; algorithmic 4*F_G + 12*F_D = %d FLOP/chunk (F_G=%d, F_D=%d)" % (fps, fg, fd)}}
        print(json.dumps(out), flush=True)
    model.close()
    rdist.barrier()


def bench_host_loop(a, net, B, n_batches, gen_updates, local, via_reader, T=100):
    """The loop a user runs (scripts/train_gan_rnn_placeholder.py:48-133): train_one_iteration over HOST numpy batches -- ragged lengths,
    padded per batch -- fed through io.prefetch (the reference's Queue of 32) and uploaded per batch (labels / lengths through
    HipEngine.upload_ready, which is what earns RSRGAN_DPIPE's guarantee); frames/s end to end, losses fetched once per iteration.
    via_reader: synthetic Kaldi arks are written to a scratch directory and read back by the bucketing reader
    (io.features.PaddedBatchReader: ark parsing, padding, length buckets) in the prefetch thread; otherwise the padded batches are
    materialised beforehand (a reader that keeps up -- the reader's own rate is reported beside it).  One untimed pass first: a
    graph segment exists per padded length T."""
    import shutil
    import tempfile
    from types import SimpleNamespace
    from rsrgan_amd import GAN_RNN, train_one_iteration
    from rsrgan_amd.io import prefetch
    from rsrgan_amd.io.features import PaddedBatchReader
    from rsrgan_amd.io.kaldi_ark import ArkWriter
    rng = np.random.default_rng(99)
    tmp = tempfile.mkdtemp(prefix="rsrgan_bench_")
    try:
        wi, wl = ArkWriter(os.path.join(tmp, "in.scp")), ArkWriter(os.path.join(tmp, "lab.scp"))
        n_utt = B * n_batches
        # lengths in {0.6, 0.7, 0.8, 0.9, 1.0} T: ragged rows, yet only a few distinct padded lengths per run (one graph segment each);
        # one length bucket (num_buckets = 1): get_padded_batch's buckets start at 200 frames, these utterances are shorter
        lens = rng.choice([int(T * f) for f in (0.6, 0.7, 0.8, 0.9, 1.0)], size=n_utt)
        for i, n in enumerate(lens):
            u = "utt%06d" % i
            wi.write_next_utt(os.path.join(tmp, "in.ark"), u, rng.standard_normal((n, 257)).astype(np.float32))
            wl.write_next_utt(os.path.join(tmp, "lab.ark"), u, rng.standard_normal((n, 40)).astype(np.float32))
        wi.close(); wl.close()
        reader = PaddedBatchReader(os.path.join(tmp, "in.scp"), os.path.join(tmp, "lab.scp"), B, num_buckets=1, shuffle=True, seed=7)
        t0 = time.perf_counter()
        batches = list(reader)
        reader_s = time.perf_counter() - t0
        batches = [b for b in batches if b[1].shape[0] == B]
        frames = sum(b[1].shape[0] * b[1].shape[1] for b in batches)
        args = SimpleNamespace(batch_size=B, input_dim=257, output_dim=40, left_context=0, right_context=0, g_type=net, keep_prob=1.0,
                               batch_norm=False, num_gpu=1, save_dir=None, l2_scale=0.0, disc_updates=1, gen_updates=gen_updates,
                               init_mse_weight=10.0, init_disc_noise_std=0.0, d_learning_rate=1e-3, g_learning_rate=8e-5)
        out = {}
        for dpipe in ("1", "0"):
            keep = os.environ.get("RSRGAN_DPIPE")
            os.environ["RSRGAN_DPIPE"] = dpipe
            try:
                model = GAN_RNN(None, args, ["gpu:%d" % local], max_frames=T, seed=4321, net_overrides=dict(flags=a.flags))
                feed = (lambda: prefetch(PaddedBatchReader(os.path.join(tmp, "in.scp"), os.path.join(tmp, "lab.scp"), B, num_buckets=1,
                                                           shuffle=True, seed=7), capacity=32)) if via_reader else (lambda: prefetch(batches, capacity=32))
                for _ in range(2):          # eager, then captured; the timed pass replays
                    train_one_iteration(None, model, len(batches), 0, feed())
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res = train_one_iteration(None, model, len(batches), 1, feed())      # (returns after the losses have reached the host)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                if not np.all(np.isfinite(res)):
                    raise RuntimeError("non-finite losses %s" % (res,))
                out["dpipe" + dpipe] = {"value": round(frames / dt, 1), "ms_per_iteration": round(dt * 1e3 / len(batches), 4)}
                del model
                torch.cuda.empty_cache()
            finally:
                if keep is None:
                    os.environ.pop("RSRGAN_DPIPE", None)
                else:
                    os.environ["RSRGAN_DPIPE"] = keep
        return {"unit": "frames/s", "batches": len(batches), "frames_per_batch_mean": round(frames / max(len(batches), 1), 1),
                "padded_lengths": sorted({int(b[1].shape[1]) for b in batches}), "RSRGAN_DPIPE=1": out["dpipe1"], "RSRGAN_DPIPE=0": out["dpipe0"],
                "reader_alone_frames_per_s": round(frames / reader_s, 1), "fed_by": "PaddedBatchReader in the prefetch thread" if via_reader
                else "pre-materialised padded batches through io.prefetch"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def spawn_ranks(n, backend):
    """Re-run this command line as n ranks of one node under torch.distributed.run (rendezvous on 127.0.0.1: the container's hostname
    may not resolve).  Fails loudly -- non-zero, nothing printed on stdout -- when the node has fewer GPUs than ranks."""
    import socket
    import subprocess
    if backend == "nccl":
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            print("bench.py: --gpus %d but %d GPU(s) visible: refusing to measure fewer devices than asked for" % (n, have), file=sys.stderr)
            return 3
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", str(max(1, _usable_cpus() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def bench_plumbing(a, rank, world):
    """--backend gloo: this script's rank plumbing on CPU, with an engine injected by the caller (RSRGAN_BENCH_ENGINE=module:factory,
    factory(batch_size, max_frames, rank) -> engine speaking rsrgan_amd/engine_hip.py's protocol).  bench.py itself never imports the
    oracle outside cpu_baseline; the product path has no CPU engine, so without an injected one this refuses."""
    import importlib
    from types import SimpleNamespace
    from rsrgan_amd import GAN_RNN, dist as rdist
    spec = os.environ.get("RSRGAN_BENCH_ENGINE")
    if not spec:
        raise SystemExit("bench.py --backend gloo: no GPU path on CPU (there is no CPU fallback); set RSRGAN_BENCH_ENGINE=module:factory")
    mod, fn = spec.split(":")
    B, T = min(a.batch, 2), min(a.frames, 4)
    engine = getattr(importlib.import_module(mod), fn)(B, T, rank)
    args = SimpleNamespace(batch_size=B, input_dim=engine.cfg.input_dim, output_dim=engine.cfg.output_dim, left_context=0, right_context=0,
                           g_type=engine.cfg.g_type, keep_prob=1.0, batch_norm=False, num_gpu=world, save_dir=None, l2_scale=0.0,
                           disc_updates=1, gen_updates=a.gen_updates, init_mse_weight=10.0, init_disc_noise_std=0.0,
                           d_learning_rate=1e-3 * world, g_learning_rate=8e-5 * world)
    model = GAN_RNN(None, args, ["cpu:%d" % rank], engine=engine, max_frames=T)
    x, lab, ln = synthetic(B, T, engine.cfg.input_dim, engine.cfg.output_dim, seed=1234 + rank)

    def step():
        model.d_step(x, lab, ln, sync=False, gather=False)
        out = None
        for i in range(a.gen_updates):
            out = model.g_step(x, lab, ln, reuse_g_forward=(i == 0), sync=False, gather=False)
        return out
    for _ in range(a.warmup):
        step()
    rdist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        last = step()
    rdist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t[0])
    if rank == 0:
        print(json.dumps({"metric": "GAN train frames/sec (G+D step), 257-dim LPS->40-dim MFCC", "value": round(B * T * world * a.steps / dt, 1),
                          "unit": "frames/s", "n_gpus": world, "rccl_ranks": rdist.world_size(), "backend": "gloo", "steps": a.steps,
                          "warmup": a.warmup, "ms_per_step": round(dt * 1e3 / a.steps, 4), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "n/a", "data": "synthetic",
                          "config": {"workload": "PLUMBING TEST ONLY: injected CPU engine %s, B=%d T=%d per rank; no GPU, not a measurement" % (spec, B, T),
                                     "global_batch": B * world, "parallelism": "dp%d" % world,
                                     "losses_last_step": [round(float(v), 6) for v in last.mean(0)]},
                          "roofline": None}), flush=True)
    rdist.barrier()


def main():
    # RSRGAN_DPIPE=1 is the default of the Python host layer (rsrgan_amd/engine_hip.py: HipEngine hands every D-run's labels and lengths
    # through upload_ready, which is the guarantee the switch asks for): D(real) of the next D-run runs beside the previous G-run's
    # weight-gradient GEMMs and the D-run itself is one launch less on the chain.  Set here too so that the line's
    # "d_pipe" field and the RSRGAN_DPIPE=0 variant read the same variable.  RSRGAN_DPIPE=0 to compare.
    os.environ.setdefault("RSRGAN_DPIPE", "1")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--net", default="lstm", choices=["lstm", "res_lstm_l", "dnn_gan", "dnn_trainer", "baseline_named", "rced", "segan"],
                    help="dnn_gan = the frame-level GAN of models/gan.py (SURVEY 8f-1): --batch frames per step, T ignored; "
                         "baseline_named = BASELINE.json's wording: 2-layer 512-unit LSTM (no projection, SURVEY 8d-iii) + DNN D")
    ap.add_argument("--d-type", default="lstm", choices=["lstm", "dnn"],
                    help="dnn = models/discriminator_dnn.py as the D of the sequence model (BASELINE.json's wording)")
    ap.add_argument("--gen-updates", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the BASELINE.json-named network variant at N=1")
    ap.add_argument("--no-hbm-activity", action="store_true", help="skip the ~4 s of extra steps under the rocm-smi memory-activity sampler")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the extra event-bracketed step (PMC passes count bytes per step)")
    ap.add_argument("--flags", type=int, default=int(os.environ.get("RSRGAN_FLAGS", "3")),
                    help="library schedule flags: 1 = wavefront, 2 = hipGraph replay, 4 = side-stream GEMM overlap (include/rsrgan.h)")
    ap.add_argument("--batch-norm", action="store_true", help="--net dnn_gan / dnn_trainer / rced: batch_norm(renorm=True) on the hidden layers "
                                                              "(run_gan_dnn.sh:134, run_dnn.sh:134)")
    ap.add_argument("--rced-gan", action="store_true", help="--net rced: 1 D + 1 G step with discriminator_dnn instead of the supervised trainer")
    ap.add_argument("--rced-width", type=int, default=40, help="--net rced: frame width (run_dnn.sh:137 uses 40-dim MFCC input)")
    ap.add_argument("--segan-len", type=int, default=16384, help="--net segan: samples per chunk (BASELINE.json configs[4]: 16384; --batch 32)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling (SURVEY 8d): --batch is the GLOBAL batch, split evenly over the ranks (default: weak, "
                         "--batch per GPU as the reference defines batch_size per tower)")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)      # internal: net,B,T,threads -> one timed CPU-baseline leg
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the product).  gloo = no GPU: only the rank plumbing of this script (spawn, rendezvous, "
                         "barriers, MAX over ranks, the one JSON line) runs, on an engine the CALLER injects through "
                         "RSRGAN_BENCH_ENGINE=module:factory -- the tests' CPU stand-in; nothing is measured that means anything")
    a = ap.parse_args()
    if a.cpu_worker:
        net, B, T, th = a.cpu_worker.split(",")
        n, dt = _cpu_run(net, int(B), int(T), int(th), 5.0, 2)
        print(json.dumps({"steps": n, "seconds": dt}), flush=True)
        return

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU, models/gan_rnn_placeholder.py:152-175's
        # tower loop; scripts/train_gan_rnn_placeholder.py:414-425,458-461) and hand their exit code on
        sys.exit(spawn_ranks(a.gpus, a.backend))

    from rsrgan_amd import GAN_RNN, dist as rdist
    rank, local, world = rdist.init_from_env(a.backend)
    if world != max(a.gpus, 1):
        # never a silent one-GPU number under an N-GPU label (or the reverse)
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch N ranks (torch.distributed.run --nproc-per-node N) with --gpus N, "
                         "or run `python bench.py --gpus N` alone, which spawns them" % (a.gpus, world))
    if a.backend == "gloo":
        return bench_plumbing(a, rank, world)
    if torch.cuda.device_count() <= local:
        raise SystemExit("bench.py: rank %d needs GPU %d, %d visible" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if a.strong:
        if a.batch % world:
            raise SystemExit("--strong: global batch %d not divisible by %d ranks" % (a.batch, world))
        a.batch //= world

    from types import SimpleNamespace
    if a.net in ("dnn_gan", "dnn_trainer"):
        return bench_dnn_gan(a, rank, local, world, dev)
    if a.net == "rced":
        return bench_rced(a, rank, local, world, dev)
    if a.net == "segan":
        return bench_segan(a, rank, local, world, dev)
    res = measure_sequence(a, a.net, a.d_type, a.batch, a.frames, a.steps, a.warmup, rank, local, world, dev)
    model, g_type, dt, dev_ms, losses, B, T = res["model"], res["g_type"], res["dt"], res["dev_ms"], res["losses"], a.batch, a.frames
    if a.net == "baseline_named":
        a.d_type = "dnn"
    if rank == 0:
        c = model.engine.cfg
        fpf, fg, fd = flop_per_frame(257, 40, g_type, c.g_layers, c.g_cells, c.g_proj, c.d_layers, c.d_cells, c.d_proj)
        if a.d_type == "dnn":      # per-frame FC stack 40 -> d_layers x d_cells -> 1
            fd = 2 * (40 * c.d_cells + (c.d_layers - 1) * c.d_cells * c.d_cells + c.d_cells)
            fpf = 3 * fg + 8 * fd
        if a.gen_updates != 1:
            fpf = None
        frames = B * T * world * a.steps
        value = frames / dt
        step_dev_s = dev_ms * 1e-3 / a.steps
        roof = None
        traffic = None
        k_traffic = None
        k_rocprof_us = None
        headline = a.net == "lstm" and a.d_type == "lstm" and (B, T, a.gen_updates) == (64, 100, 1)
        # the newest committed set of final profiles (profiles/r<N>_final_*)
        import glob, re
        rounds = sorted({int(m.group(1)) for f in glob.glob(os.path.join(ROOT, "profiles", "r*_final_traffic.json"))
                         for m in [re.search(r"r(\d+)_final_traffic", f)] if m})
        ptag = "r%d" % rounds[-1] if rounds else "r4"
        tf_path = os.path.join(ROOT, "profiles", "%s_final_traffic.json" % ptag)
        if headline and os.path.exists(tf_path):
            # HBM-side bytes from the committed PMC passes of this workload (tools/traffic.sh: separate FETCH_SIZE / WRITE_SIZE
            # passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); refused when the kernels have changed
            # since (recorded ms/step more than 10 % away from this run)
            tj = json.load(open(tf_path))
            rec = tj.get("ms_per_step")
            if rec and abs(rec - step_dev_s * 1e3) <= 0.10 * step_dev_s * 1e3:
                traffic = tj.get("hbm_bytes_per_step")
                f = [v for v in tj.get("top_fetch", []) if res["prof"][0] in v[0]]
                w = [v for v in tj.get("top_write", []) if res["prof"][0] in v[0]]
                if f and w and f[0][2] == w[0][2]:
                    k_traffic = int((2 * f[0][1] + w[0][1]) * 1024 / f[0][2])
        cs_path = os.path.join(ROOT, "profiles", "%s_final_rocprofv3_kernel_stats.csv" % ptag)
        if headline and os.path.exists(cs_path):
            import csv
            for row in csv.DictReader(open(cs_path)):
                if res["prof"][0] in row["Name"]:
                    k_rocprof_us = round(float(row["AverageNs"]) / 1e3, 3); break
        if fpf:
            ach = fpf * B * T / step_dev_s / 1e12          # per GPU, HIP-event time of the whole step's launches
            roof = {"bound": "mfma", "achieved": round(ach, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                    "scope": "all launches of one (1D+1G) step on one GPU; algorithmic %d FLOP/frame x %d frames "
                             "(SURVEY 8d: 3*F_G+8*F_D, F_G=%d, F_D=%d) / HIP-event step time; traffic = HBM-side bytes "
                             "per step from profiles/%s_final_traffic.json (null when that file is absent or stale)" % (fpf, B * T, fg, fd, ptag)}
            ALG_BYTES = 1.3e9          # ~1.1 GB activation stash write + read, 164 MB Adam, 24 MB parameters, 7.6 MB inputs
            ratio = {"algorithmic_bytes_per_step": int(ALG_BYTES)}
            if traffic:
                ratio["pmc"] = round(traffic / ALG_BYTES, 2)
            if res.get("hbm"):
                roof["traffic_is"] = "L2 fabric requests of serialised cold-L2 dispatches incl. Infinity-Cache hits (PMC): an upper bound"
                roof["hbm_activity"] = res["hbm"]
                ratio["memory_controllers"] = round(res["hbm"]["hbm_bytes_per_step_estimate"] / ALG_BYTES, 2)
            roof["traffic_ratio"] = ratio
            if res.get("chain"):
                # the third bound: the recurrence is a chain of dependent launches; each costs at least a kernel boundary
                # plus one dependent operand round trip, whatever its FLOPs
                n_c, us0, us1 = res["chain"]
                roof["latency_bound_ms"] = round(n_c * us1 * 1e-3, 3)
                roof["latency_bound"] = {"chain_launches_per_step": int(n_c), "floor_us_per_launch": round(us1, 3),
                                         "kernel_boundary_us": round(us0, 3),
                                         "how": "launches of the recurrence kernels (gates, projection, backward A / B / B-reduce) issued "
                                                "in one (1D+1G) step x the measured cost of one launch in a replayed hipGraph of 400 dependent "
                                                "256-workgroup launches that each read 1 KB per wave of their predecessor's output "
                                                "(rsrgan_op_launch_floor mode 1; mode 0 = empty kernels = the boundary alone)"}
            k_name, n_l, us_l, fl_l = res["prof"]
            if n_l:
                # the step is GPU-bound (sum of rocprof kernel durations = wall, profiles/r2_gap_summary.txt), so the kernel's
                # duration is what rocprofv3 reports; the live HIP-event bracket also contains the two event records and is
                # an upper bound -- both are given, the roofline fraction uses the live one measured in THIS run
                k_ach = fl_l / (us_l * 1e-6) / 1e12
                roof["dominant_kernel"] = {
                    "name": k_name, "launches_per_step": n_l, "avg_us": round(us_l / n_l, 3),
                    "rocprofv3_avg_us": k_rocprof_us,
                    "algorithmic_flop_per_launch": round(fl_l / n_l), "achieved": round(k_ach, 3), "unit": "TFLOP/s",
                    "frac": round(k_ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": k_traffic,
                    "frac_at_rocprofv3_duration": (round(fl_l / n_l / (k_rocprof_us * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)
                                                   if k_rocprof_us else None),
                    "how": "the kernel class with the largest total duration in the step: k_glstm_bwd (csrc/gpersist.hip: the generator's "
                           "whole BPTT as ONE persistent launch; algorithmic FLOP = every layer's state-gradient product and dh = dm.Wp^T "
                           "+ the input-gradient product above layer 0; round 5: the launch is k_glstm_bwd_dt, which also carries the "
                           "discriminator's trailing BPTT of the G-run -- its data-gradient products and dy.W_out^T are counted), k_glstm_fwd (its forward recurrence: recurrent product, projection, "
                           "input product above layer 0) or, with RSRGAN_GPERSIST=0, the k_fwd_gates launches of the wavefront.  avg_us: "
                           "every launch of one extra step bracketed by HIP events on its stream (rsrgan_profile_begin / read_kind / read; "
                           "includes the event records and the memset that arms the launch's hop-2 slots, an upper bound); "
                           "rocprofv3_avg_us: AverageNs of the same kernel in the committed profiles/%s_final_rocprofv3_kernel_stats.csv "
                           "of this command" % ptag}
        out = {"metric": "GAN train frames/sec (G+D step), 257-dim LPS->40-dim MFCC", "value": round(value, 1),
               "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(dt * 1e3 / a.steps, 4), "ms_per_step_median": round(res["med_ms"], 4),
               "higher_is_better": True, "scaling": "strong" if a.strong else "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic", "rccl_ranks": rdist.world_size(),
               "arith": ARITH_NOTE if os.environ.get("RSRGAN_GP_TAGS", "1") != "0" else "fp32 MFMA / fp32 accumulate, untagged hand-offs (RSRGAN_GP_TAGS=0)",
               "parity_margin": parity_margin(g_type, B, T),
               "config": {"workload": "gan_rnn_placeholder 1D+%dG step, G=%s(%dx%d/p%d)+D=%s(%dx%d/p%d), B=%d/GPU T=%d, "
                                      "257->40" % (a.gen_updates, g_type, c.g_layers, c.g_cells, c.g_proj, a.d_type, c.d_layers,
                                                   c.d_cells, c.d_proj, B, T),
                          "schedule_flags": a.flags, "d_pipe": os.environ.get("RSRGAN_DPIPE", "0"), "global_batch": B * world, "frames_per_step": B * T * world, "parallelism": "dp%d" % world,
                          "losses_last_step": [round(float(v), 6) for v in losses]},
               "roofline": roof}
        if res.get("buckets"):
            out["allreduce"] = dict(res["buckets"], how="rank 0, generator gradients of one extra step: all-reduce of bucket i on the "
                                    "communication stream vs the moment the compute stream finished the last weight-gradient GEMM; "
                                    "exposed_ms is what the step pays for communication (HipEngine.bucket_report)")
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.net if a.net in ("lstm", "res_lstm_l") else "lstm", B, T)
    if world == 1 and a.net == "lstm" and a.d_type == "lstm" and not a.no_variants:
        # BASELINE.json words its configs as "2-layer 512-unit LSTM generator + DNN discriminator" (no such network exists
        # in the reference); the same kernels run it, reported beside the reference-true headline.
        del model, res
        torch.cuda.empty_cache()
        v = measure_sequence(a, "baseline_named", "dnn", B, T, max(10, a.steps // 2), 5, rank, local, world, dev)      # (variants: >= 10 timed steps after 5)
        c2 = v["model"].engine.cfg
        f2, fg2, fd2 = flop_per_frame(257, 40, "res_lstm_base", 2, 512, 0, c2.d_layers, c2.d_cells, 0)
        fd2 = 2 * (40 * c2.d_cells + (c2.d_layers - 1) * c2.d_cells * c2.d_cells + c2.d_cells)
        f2 = 3 * fg2 + 8 * fd2
        n2 = max(10, a.steps // 2)
        out["variants"] = [{"workload": "BASELINE.json-named: G=2x512 LSTM (num_proj=None) + D=discriminator_dnn(4x1024), B=%d T=%d" % (B, T),
                            "value": round(B * T * n2 / v["dt"], 1), "unit": "frames/s", "ms_per_step": round(v["dt"] * 1e3 / n2, 4),
                            "roofline_frac": round(f2 * B * T / (v["dev_ms"] * 1e-3 / n2) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                            "flop_per_frame": f2}]
        # the network the shipped recipe selects (run_gan_rnn_placeholder.sh:124,149: --g_type res_lstm_l = 4 x LSTMP(760, p257) with the
        # running residual sum, models/res_lstm_l.py:101-194): 3*F_G + 8*F_D with F_G = 14 083 600
        del v
        torch.cuda.empty_cache()
        vr = measure_sequence(a, "res_lstm_l", "lstm", B, T, n2, 5, rank, local, world, dev)
        cr = vr["model"].engine.cfg
        fr, fgr, fdr = flop_per_frame(257, 40, vr["g_type"], cr.g_layers, cr.g_cells, cr.g_proj, cr.d_layers, cr.d_cells, cr.d_proj)
        out["variants"].append({"workload": "shipped network: G=res_lstm_l(4x760/p257) + D=lstm(2x256/p40), 1D+1G, B=%d T=%d" % (B, T),
                                "value": round(B * T * n2 / vr["dt"], 1), "unit": "frames/s", "ms_per_step": round(vr["dt"] * 1e3 / n2, 4),
                                "roofline_frac": round(fr * B * T / (vr["dev_ms"] * 1e-3 / n2) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                                "flop_per_frame": fr, "F_G": fgr, "F_D": fdr})
        del vr
        torch.cuda.empty_cache()
        # the shipped schedule: 1 D-run + 2 G-runs per batch (run_gan_rnn_placeholder.sh:129-130), reference-true networks
        a2 = argparse.Namespace(**vars(a)); a2.gen_updates = 2
        v2 = measure_sequence(a2, "lstm", "lstm", B, T, n2, 5, rank, local, world, dev)
        out["variants"].append({"workload": "shipped schedule 1D+2G per batch, reference-true networks, B=%d T=%d" % (B, T),
                                "value": round(B * T * n2 / v2["dt"], 1), "unit": "frames/s", "ms_per_step": round(v2["dt"] * 1e3 / n2, 4)})
        del v2
        torch.cuda.empty_cache()
        # the headline workload WITHOUT the caller's guarantee behind RSRGAN_DPIPE (every buffer of rsrgan_d_step read in stream order:
        # the D-run's two discriminator calls stacked behind the generator's forward launch) -- what a caller that uploads a batch
        # right before the call gets
        if os.environ.get("RSRGAN_DPIPE", "0") != "0":
            keep = os.environ["RSRGAN_DPIPE"]; os.environ["RSRGAN_DPIPE"] = "0"
            try:
                v0 = measure_sequence(a, "lstm", "lstm", B, T, max(10, a.steps // 2), 4, rank, local, world, dev)
                n0 = max(10, a.steps // 2)
                out["variants"].append({"workload": "the headline step with RSRGAN_DPIPE=0 (no guarantee about the caller's buffers), B=%d T=%d" % (B, T),
                                        "value": round(B * T * n0 / v0["dt"], 1), "unit": "frames/s", "ms_per_step": round(v0["dt"] * 1e3 / n0, 4),
                                        "roofline_frac": round(fpf * B * T / (v0["dev_ms"] * 1e-3 / n0) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)})
                del v0
            finally:
                os.environ["RSRGAN_DPIPE"] = keep
            torch.cuda.empty_cache()
        # the shipped RECIPE as a whole (run_gan_rnn_placeholder.sh:124,126,129-130): res_lstm_l, batch_size 8, 1 D-run + 2 G-runs -- the
        # batch is padded to one 32-row group of the persistent kernels (csrc/model.h Bt); and the same network at BASELINE configs[1]'s batch
        for (bb, gu, tag) in ((8, 2, "shipped recipe: G=res_lstm_l, batch_size 8, 1D+2G"), (32, 1, "shipped network at B=32 (one row group: persistent launches), 1D+1G")):
            a5 = argparse.Namespace(**vars(a)); a5.gen_updates = gu
            v5 = measure_sequence(a5, "res_lstm_l", "lstm", bb, T, n2, 5, rank, local, world, dev)      # (warm-up steps: a segment is captured at its second use, replayed from the third)
            out["variants"].append({"workload": "%s, T=%d" % (tag, T), "value": round(bb * T * n2 / v5["dt"], 1), "unit": "frames/s",
                                    "ms_per_step": round(v5["dt"] * 1e3 / n2, 4)})
            del v5
            torch.cuda.empty_cache()
        # the loop a user runs: train_one_iteration fed from host batches (ragged, padded per batch) through io.prefetch + upload_ready
        for (net_h, bb, gu, nb, rd, tag) in (("res_lstm_l", 8, 2, 50, True, "shipped recipe (res_lstm_l, batch_size 8, 1D+2G), 50 ragged host batches read from Kaldi arks by the bucketing reader"),
                                             ("lstm", B, 1, 24, False, "headline networks, B=%d, 1D+1G, 24 ragged host batches" % B)):
            try:
                h = bench_host_loop(a, net_h, bb, nb, gu, local, rd, T)
                out["variants"].append(dict({"workload": "train_one_iteration end to end: " + tag}, **h))
            except Exception as e:      # never lose the headline line to a variant
                out["variants"].append({"workload": "train_one_iteration end to end: " + tag, "error": str(e)[:200]})
            torch.cuda.empty_cache()
        try:           # BASELINE.json configs[3]: R-CED (257 x 11) + discriminator_dnn, N = 6400 frames (bench.py --net rced --rced-gan)
            import contextlib, io
            a3 = argparse.Namespace(**vars(a)); a3.net = "rced"; a3.rced_gan = True; a3.rced_width = 257; a3.batch = 6400; a3.steps = 5; a3.warmup = 2
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                bench_rced(a3, rank, local, world, dev)
            r3 = json.loads(buf.getvalue().strip().splitlines()[-1])
            out["variants"].append({"workload": r3["config"]["workload"], "value": r3["value"], "unit": "frames/s",
                                    "ms_per_step": r3["ms_per_step"], "roofline_frac": r3["roofline"]["frac"],
                                    "roofline_frac_incl_padding": r3["roofline"]["frac_incl_padding"]})
        except Exception as e:          # never lose the headline line to a variant
            out["variants"].append({"workload": "R-CED + discriminator_dnn (configs[3])", "error": str(e)[:200]})
        try:           # BASELINE.json configs[4]: SEGAN-style conv G/D, 16384-sample chunks, B = 32 (bench.py --net segan --batch 32)
            import contextlib, io
            a4 = argparse.Namespace(**vars(a)); a4.net = "segan"; a4.batch = 32; a4.segan_len = 16384; a4.steps = 10; a4.warmup = 3
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                bench_segan(a4, rank, local, world, dev)
            r4 = json.loads(buf.getvalue().strip().splitlines()[-1])
            out["variants"].append({"workload": r4["config"]["workload"], "value": r4["value"], "unit": r4["unit"],
                                    "ms_per_step": r4["ms_per_step"], "roofline_frac": r4["roofline"]["frac"]})
        except Exception as e:
            out["variants"].append({"workload": "SEGAN-style conv G/D (configs[4])", "error": str(e)[:200]})
    if rank == 0:
        print(json.dumps(out), flush=True)
    rdist.barrier()


if __name__ == "__main__":
    main()
