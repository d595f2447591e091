"""Worker of tests/test_ingest_ranks.py (gloo, no GPU): every rank ingests its share of the read files by bytes
(shannon_amd.distributed.ingest_rank_slice) and writes its code matrices + what it looked at; also the tensor all-gather of
arrays that replaced the pickled candidate gather (exchange.all_gather_arrays)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
from shannon_amd import distributed, exchange


def main():
    out, paths = sys.argv[1], sys.argv[2:]
    dist.init_process_group("gloo")
    rank, W = dist.get_rank(), dist.get_world_size()
    stats = {}
    exchange.stats_reset()
    sl = distributed.ingest_rank_slice(paths, rank, W, None, torch.device("cpu"), stats=stats)
    if sl is None:
        json.dump({"declined": True}, open("%s.rank%d.json" % (out, rank), "w"))
    else:
        mats, n = sl
        np.savez("%s.rank%d.npz" % (out, rank), **{"m%d" % i: m for i, m in enumerate(mats)})
        stats["n"] = n
        json.dump(stats, open("%s.rank%d.json" % (out, rank), "w"))
    # arrays of every rank to every rank as tensors: sizes differ per rank, one of them empty on rank 1
    rng = np.random.default_rng(rank)
    mine = (rng.integers(0, 1 << 40, 5 + 3 * rank).astype(np.int64), rng.integers(0, 1 << 60, 5 + 3 * rank, dtype=np.uint64),
            np.arange(0 if rank == 1 else 100 * (rank + 1), dtype=np.uint64), rng.integers(0, 4, 1000 * rank + 7).astype(np.uint8))
    got = exchange.all_gather_arrays(mine, torch.device("cpu"), None, "test arrays")
    want = [None] * W
    dist.all_gather_object(want, mine)
    ok = all(len(g) == len(w) and all(a.dtype == b.dtype and np.array_equal(a, b) for a, b in zip(g, w)) for g, w in zip(got, want))
    snap = exchange.stats_snapshot()
    json.dump({"arrays_equal": bool(ok), "stats": {k: v for k, v in snap.items()}}, open("%s.gather%d.json" % (out, rank), "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
