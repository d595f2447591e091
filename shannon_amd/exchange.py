"""k1-mer bucket exchange across the GPUs of one node: the only collective of the
path.  Each rank counts its own read shard, shards its (canonical key, count) table by
owner = fmix64(key ^ SALT) mod world, and one all-to-all(v) over RCCL/xGMI (torch.distributed
backend "nccl" on ROCm) moves every pair to its owner; the owner reduces by key.

The reference has no collective at all (single node, multiprocessing + files); this replaces
nothing in it -- it is what lets `jellyfish count` (shannon.py:439) shard across 8 GPUs.
"""
import numpy as np
import torch
import torch.distributed as dist

SHARD_SALT = 0xA24BAED4963EE407
M64 = (1 << 64) - 1


def fmix64_np(x):
    x = x.astype(np.uint64).copy()
    x ^= x >> np.uint64(33)
    x *= np.uint64(0xff51afd7ed558ccd)
    x ^= x >> np.uint64(33)
    x *= np.uint64(0xc4ceb9fe1a85ec53)
    x ^= x >> np.uint64(33)
    return x


def owner_of(keys, world):
    """Host mirror of owner_of() in csrc/count.hip (used by tests and by the host-side planner)."""
    with np.errstate(over="ignore"):
        return (fmix64_np(np.asarray(keys, dtype=np.uint64) ^ np.uint64(SHARD_SALT)) % np.uint64(world)).astype(np.int64)


def coll_device(dev, group=None):
    """Where the tensors of a collective live: on their own device with RCCL ("nccl"), in host memory with gloo
    (CPU tests, and GPU tests in which several ranks share one GPU -- RCCL refuses two ranks on one device)."""
    return torch.device("cpu") if dist.get_backend(group) == "gloo" else torch.device(dev)


def all_to_all_pairs(keys, counts, send_counts, group=None):
    """keys (int64) / counts (int32) tensors grouped by destination rank with `send_counts`
    entries each.  Returns (recv_keys, recv_counts, recv_counts_per_rank)."""
    world = dist.get_world_size(group)
    dev = keys.device
    cdev = coll_device(dev, group)
    sc = torch.as_tensor(np.asarray(send_counts, dtype=np.int64), device=cdev)
    rc = torch.empty(world, dtype=torch.int64, device=cdev)
    dist.all_to_all_single(rc, sc, group=group)
    rcl = [int(v) for v in rc.cpu().tolist()]
    scl = [int(v) for v in np.asarray(send_counts).tolist()]
    n_in = sum(rcl)
    rk = torch.empty(n_in, dtype=torch.int64, device=cdev)
    rcn = torch.empty(n_in, dtype=torch.int32, device=cdev)
    dist.all_to_all_single(rk, keys[:sum(scl)].to(cdev), rcl, scl, group=group)
    dist.all_to_all_single(rcn, counts[:sum(scl)].to(cdev), rcl, scl, group=group)
    return rk.to(dev), rcn.to(dev), rcl


def exchange_table(ctx, table, group=None):
    """Device path: shard `table` (device.Table, canonical keys) by owner, all-to-all, reduce by key.
    Returns the owned device.Table."""
    from . import device
    world = dist.get_world_size(group)
    n = len(table)
    dk = torch.empty(max(n, 1), dtype=torch.int64, device="cuda")
    dc = torch.empty(max(n, 1), dtype=torch.int32, device="cuda")
    per = table.shard(world, dk.data_ptr(), dc.data_ptr())
    rk, rcn, _ = all_to_all_pairs(dk, dc, per, group)
    torch.cuda.synchronize()
    return device.Table.from_pairs(ctx, rk.data_ptr(), rcn.data_ptr(), rk.numel(), table.k, table.canonical)
