"""a rank that receives no component: the extension on an empty table"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from shannon_amd import device, extension_correction as ec
ctx = device.Context(0)
k = torch.zeros(1, dtype=torch.int64, device="cuda"); c = torch.zeros(1, dtype=torch.int32, device="cuda")
t = device.Table.from_pairs(ctx, k.data_ptr(), c.data_ptr(), 0, 26, True)
print("len", len(t))
e = ec.Extension(ctx, t, 3)
print("walks", e.n_walks)
e.close()


class G(object):
    world, rank = 2, 0

    @staticmethod
    def all_gather(obj):
        return [obj, obj]

    @staticmethod
    def all_reduce_max(v):
        return int(v)


for big in ("0", "1"):
    os.environ["SHN_CONTIG_GPU"] = big
    res = ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False, shard=None, gather=G, table_size=0)
    print("big", big, "contigs", len(res.contigs))
print("OK")
