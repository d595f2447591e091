"""`python bench.py --gpus N` with no WORLD_SIZE in the environment must start the N ranks itself (fresh child processes through
torch.distributed.run from a parent that made no GPU call), relay rank 0's JSON line and nothing else on stdout, and fail when a
rank fails.  CPU test: SHN_BENCH_LAUNCH_PROBE makes the ranks meet over gloo and report what the launch resolved to (no GPU)."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(args, probe="1", extra_env=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["SHN_BENCH_LAUNCH_PROBE"] = probe
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)


def test_gpus_n_starts_n_ranks_on_the_configs3_batch():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines                              # stdout carries the JSON line only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2
    assert d["scaling"] == "strong" and d["config"] == "2" and d["K"] == 25 and d["genes"] == 20000
    assert d["reads_of_the_job"] == 100_000_000 and d["reads_of_rank0"] == 50_000_000 and d["first_chunk"] == 0
    assert d["steps"] == 3 and d["warmup"] == 1


def test_three_ranks_weak_and_the_k31_slice():
    r = _run(["--gpus", "3", "--scaling", "weak"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = json.loads(r.stdout.decode().strip())
    assert d["n_gpus"] == 3 and d["rccl_ranks"] == 3 and d["scaling"] == "weak" and d["config"] == "1" and d["reads_of_rank0"] == 10_000_000
    r = _run(["--gpus", "1", "--config", "4s"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = json.loads(r.stdout.decode().strip())
    assert d["n_gpus"] == 1 and d["K"] == 31 and d["genes"] == 4000 and d["reads_of_the_job"] == 100_000_000


def test_a_failing_rank_fails_the_launch():
    r = _run(["--gpus", "2"], probe="fail")
    assert r.returncode != 0
    assert not r.stdout.decode().strip()


def test_chunks_of_the_batch_are_dealt_contiguously():
    sys.path.insert(0, ROOT)
    import bench
    total = 50_000_000
    for W in (1, 2, 3, 4, 8):
        at, pairs = 0, 0
        for r in range(W):
            lo, n = bench.chunk_range(total, W, r)
            assert lo == at
            at += (n + bench.CHUNK_PAIRS - 1) // bench.CHUNK_PAIRS
            pairs += n
        assert pairs == total
    assert bench._chunk_seed(5, 0) != bench._chunk_seed(5, 1) != bench._chunk_seed(6, 0)
