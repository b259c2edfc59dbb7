"""bench.py's rank plumbing (the reference's tower loop models/gan_rnn_placeholder.py:152-175, LR x num_gpu
scripts/train_gan_rnn_placeholder.py:458-461): `python bench.py --gpus N` must start N ranks BY ITSELF and print one JSON line with
n_gpus = N -- never a one-rank number under an N-GPU label.  Runs on CPU: gloo, an oracle-backed engine injected by this test."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env=None, timeout=300):
    e = dict(os.environ, RSRGAN_BENCH_ENGINE="tests.helpers:bench_engine", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=timeout, cwd=ROOT)


def test_bench_gpus_2_spawns_two_ranks_by_itself():
    p = _bench(["--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "1"])
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                      # rank 0 alone prints
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2, out
    assert out["config"]["global_batch"] == 2 * 2 and out["config"]["parallelism"] == "dp2"
    assert out["steps"] == 1 and out["value"] > 0


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """--gpus 4 inside a 1-rank environment that claims to be a launcher (WORLD_SIZE=1): refuse, do not measure one rank."""
    p = _bench(["--gpus", "4", "--backend", "gloo", "--steps", "1", "--warmup", "0"], env={"WORLD_SIZE": "1", "RANK": "0"})
    assert p.returncode != 0 and "--gpus 4 but WORLD_SIZE=1" in p.stderr, (p.returncode, p.stderr[-500:])
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_bench_nccl_refuses_more_ranks_than_gpus():
    """the product backend on a box with fewer GPUs than ranks (here: none): non-zero exit, no JSON line"""
    p = _bench(["--gpus", "8", "--steps", "1", "--warmup", "0"], env={"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    assert p.returncode != 0 and "refusing" in p.stderr, (p.returncode, p.stderr[-500:])
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
