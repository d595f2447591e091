"""Worker for tests/test_distributed_gpu.py: world_size > 1 with every rank on cuda:0.  The per-rank compute is the
product's (GpuOps: HIP kernels through the C ABI); the collectives run on gloo through host memory, because RCCL
does not accept two ranks on one device (exchange.coll_device)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist


def main():
    paired, n_genes, seed, n_pairs, out = sys.argv[1] == "1", int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    ss = len(sys.argv) > 6 and sys.argv[6] == "ss"             # -s / --strand_specific
    from shannon_amd import device, synth, distributed, kmers_for_component as kfc
    dist.init_process_group("gloo")
    rank, W = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    (q1, q2), _ = synth.make_dataset(n_pairs, n_genes, seed=seed)
    if not paired:
        q1, q2 = np.concatenate([q1, q2]), None
    n = len(q1)
    lo, hi = rank * n // W, (rank + 1) * n // W
    q1 = q1[lo:hi]
    q2 = q2[lo:hi] if paired else None
    ctx = device.Context(0)
    d1 = device.Reads.from_codes(ctx, q1)
    d2 = device.Reads.from_codes(ctx, q2) if paired else None
    ops = distributed.GpuOps(ctx, d1, d2, kfc.ReadStore(q1, q2), 25)
    T = {}
    res = distributed.assemble_distributed(ops, 25, 500, "t", 1, double_stranded=not ss, timings=T)
    sizes = [None] * W
    dist.all_gather_object(sizes, getattr(ops, "component_table_sizes", None))
    digs = [None] * W
    dist.all_gather_object(digs, getattr(ops, "digests", None))
    if rank == 0:
        json.dump({"partitions": dict(res["partitions"]), "final": res["final"], "contigs": res["contigs"], "timings": T,
                   "n_k1mers": res["n_k1mers"], "table_sizes": sizes, "digests": digs}, open(out, "w"))
    dist.barrier()
    d1.close()
    if d2 is not None:
        d2.close()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
