import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
import bench
from shannon_amd import device, distributed, exchange, kmers_for_component as kfc
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
r1, r2 = bench.gen_reads(5_000_000, 20240501, 1, dev)
ctx = device.Context(0)
d1, d2 = device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)
store = kfc.ReadStore(r1, r2)
ops = distributed.GpuOps(ctx, d1, d2, store, 25)
def T(name, f):
    torch.cuda.synchronize(); t = time.time(); r = f(); torch.cuda.synchronize(); print("%-28s %.1f ms" % (name, 1e3 * (time.time() - t))); return r
for rep in range(2):
    keys, counts, send = T("local_pairs", lambda: ops.local_pairs(1))
    rk, rc, _ = T("all_to_all_pairs", lambda: exchange.all_to_all_pairs(keys, counts, send))
    ok, oc = T("reduce_pairs", lambda: ops.reduce_pairs(rk, rc))
    gk, _ = T("allgather keys", lambda: distributed._all_gather_var(ok))
    gc, _ = T("allgather counts", lambda: distributed._all_gather_var(oc))
    table = T("table_from_pairs", lambda: ops.table_from_pairs(gk, gc))
    res = T("extension", lambda: ops.extension(table, 500))
    part = T("route", lambda: ops.route(res, 25, 500, None))
    print()
dist.destroy_process_group()
