"""The N-rank path at a size where its machinery is at work (2.4 M reads, 400 genes: ~20 M k1-mers, tens of thousands of components, the
large ones dealt by size, the replicated GPU contig stage): `bench.py --gpus N` with the ranks sharing the one GPU (collectives over
gloo) must produce the transcripts of the one-GPU pipeline on the same batch -- on the default path (components labelled on owner
shards) and on the replicated-table path."""
import json, os, subprocess, sys
import pytest
from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(n, env_extra):
    env = dict(os.environ, SHN_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", **env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--scaling", "strong", "--genes", "400", "--reads", "2400000",
                        "--K", "25", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--overlap-steps", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_two_and_three_ranks_give_the_transcripts_of_one_gpu():
    one = _bench(1, {})
    sha = one["config"]["transcripts_sha256_16"]
    assert one["n_gpus"] == 1 and one["config"]["transcripts"] > 400
    for n, env in ((2, {}), (3, {}), (2, {"SHN_OWNER_LABELS": "0"})):
        got = _bench(n, env)
        assert got["n_gpus"] == n and got["config"]["rccl_ranks"] == n
        assert got["config"]["transcripts_sha256_16"] == sha, (n, env, got["config"]["transcripts"], one["config"]["transcripts"])
        stages = got["config"]["host_stage_seconds_per_step"]
        assert ("x:component exchange" in stages) == (env.get("SHN_OWNER_LABELS") != "0")
