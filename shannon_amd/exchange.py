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

# per-collective tallies of this rank (bench.py prints them per step): name -> calls, bytes sent to / received from OTHER ranks,
# seconds inside the calls (waiting for the slowest rank included)
_STATS = {}


def stats_reset():
    _STATS.clear()


def stats_snapshot():
    return {k: dict(v) for k, v in _STATS.items()}


def note(name, sent, received, seconds, calls=1):
    if not name:
        return
    e = _STATS.setdefault(name, {"calls": 0, "bytes_sent": 0, "bytes_received": 0, "seconds": 0.0})
    e["calls"] += calls
    e["bytes_sent"] += int(sent)
    e["bytes_received"] += int(received)
    e["seconds"] += float(seconds)


def coll_stats():
    """SHN_COLL_STATS=1: exact byte tallies (objects are pickled a second time to be measured) and a device synchronize after
    every all-to-all so that its seconds are the collective's own.  Off (default): seconds of the host call only, bytes from the
    arrays' sizes -- nothing extra inside a timed step."""
    import os
    return os.environ.get("SHN_COLL_STATS", "0") == "1"


def _payload_bytes(obj):
    """bytes of the numpy / bytes / str payload of a control object (what dominates its pickle), without pickling it"""
    if isinstance(obj, np.ndarray):
        return obj.nbytes
    if isinstance(obj, (bytes, bytearray, str)):
        return len(obj)
    if isinstance(obj, dict):
        return sum(_payload_bytes(v) for v in obj.values())
    if isinstance(obj, (list, tuple)):
        return sum(_payload_bytes(v) for v in obj)
    return 8


def all_gather_object(obj, group=None, name="objects"):
    """dist.all_gather_object with its size tallied (small control payloads: contigs, FASTA texts, stage dicts); the seconds are
    taken as the collective returns, the bytes from the payload arrays (pickled sizes only under SHN_COLL_STATS=1)"""
    import time
    W = dist.get_world_size(group)
    parts = [None] * W
    t0 = time.time()
    dist.all_gather_object(parts, obj, group=group)
    seconds = time.time() - t0
    if W > 1:
        me = dist.get_rank(group)
        if coll_stats():
            import pickle
            size = lambda o: len(pickle.dumps(o, protocol=pickle.HIGHEST_PROTOCOL))
        else:
            size = _payload_bytes
        note(name, size(obj) * (W - 1), sum(size(p) for i, p in enumerate(parts) if i != me), seconds)
    return parts


def all_gather_arrays(arrays, device, group=None, name="arrays"):
    """Every rank's tuple of numpy arrays to every rank, as TENSORS (one padded all_gather per array, sizes first): what used to
    travel as pickled objects (all_gather_object) -- the accepted candidates of the sharded contig stage, 0.3 GB at BASELINE
    configs[2].  Returns a list over the ranks of tuples of arrays (dtypes and shapes[1:] as given; all ranks pass the same number of
    arrays of the same dtypes).  device: where the collective's tensors live (coll_device)."""
    import time
    W = dist.get_world_size(group)
    t0 = time.time()
    arrays = [np.ascontiguousarray(a) for a in arrays]
    sizes = torch.tensor([a.size for a in arrays], dtype=torch.int64, device=device)
    all_sizes = [torch.zeros_like(sizes) for _ in range(W)]
    dist.all_gather(all_sizes, sizes, group=group)
    all_sizes = np.stack([x.cpu().numpy() for x in all_sizes])                 # [W, n_arrays]
    out = [[] for _ in range(W)]
    sent = recv = 0
    for j, a in enumerate(arrays):
        mx = int(all_sizes[:, j].max())
        raw = a.reshape(-1).view(np.uint8)
        isz = a.dtype.itemsize
        buf = torch.zeros(max(mx * isz, 1), dtype=torch.uint8, device=device)
        if raw.size:
            buf[:raw.size] = torch.from_numpy(raw.copy()).to(device)
        parts = [torch.empty_like(buf) for _ in range(W)]
        dist.all_gather(parts, buf, group=group)
        tail = a.shape[1:]
        for r in range(W):
            n_el = int(all_sizes[r, j])
            got = parts[r][:n_el * isz].cpu().numpy().view(a.dtype)
            out[r].append(got.reshape((-1,) + tuple(tail)) if tail else got)
            recv += n_el * isz
        sent += raw.size * (W - 1)
    if W > 1:
        note(name, sent, recv, time.time() - t0)
    return [tuple(x) for x in out]


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


OWNER_M = 13
OWNER_SALT = 0x6A09E667F3BCC909


def owner_of_minimizer(keys, k, canonical, world):
    """Host mirror of shn_owner_minimizer() in csrc/common.h: the rank that owns a k-mer = hash of the order value of its minimizer
    (the m-mer, m = min(13, k), of smallest murmur3-finalised value among its (canonical) m-mers) mod world.  Tests and planners."""
    keys = np.asarray(keys, dtype=np.uint64)
    m = min(OWNER_M, int(k))
    mmask = np.uint64((1 << (2 * m)) - 1)
    best = np.full(len(keys), 0xFFFFFFFF, dtype=np.uint64)
    M32 = np.uint64(0xFFFFFFFF)
    for i in range(k - m + 1):
        f = (keys >> np.uint64(2 * (k - m - i))) & mmask
        c = f
        if canonical:
            r = np.zeros_like(f)
            x = f.copy()
            for _ in range(m):
                r = (r << np.uint64(2)) | (np.uint64(3) - (x & np.uint64(3)))
                x >>= np.uint64(2)
            c = np.minimum(f, r)
        x = c & M32
        x ^= x >> np.uint64(16); x = (x * np.uint64(0x85ebca6b)) & M32
        x ^= x >> np.uint64(13); x = (x * np.uint64(0xc2b2ae35)) & M32
        x ^= x >> np.uint64(16)
        best = np.minimum(best, x)
    with np.errstate(over="ignore"):
        return (fmix64_np(best ^ np.uint64(OWNER_SALT)) % np.uint64(world)).astype(np.int64)


def coll_device(dev, group=None):
    """Where the tensors of a collective live: on their own device with RCCL ("nccl"), in host memory with gloo
    (CPU tests, and GPU tests in which several ranks share one GPU -- RCCL refuses two ranks on one device)."""
    return torch.device("cpu") if dist.get_backend(group) == "gloo" else torch.device(dev)


def chunk_elems():
    """Elements one rank sends to one peer in a single collective call.  A 100 M-read batch has 724 M distinct k1-mers: its
    bucket exchange moves 5.8 GB of keys, and one all_to_all_single of that size came back with half of the pairs (element /
    byte counts past 2^32 inside the call) -- so every variable-size collective here goes in rounds of at most this many
    elements per (source, destination) pair.  SHN_COLL_CHUNK overrides (the CPU tests force several rounds with it)."""
    import os
    return max(1, int(os.environ.get("SHN_COLL_CHUNK", 1 << 26)))


def all_to_all_v(send, scl, rcl, cdev, group=None, name=None):
    """all-to-all(v) of a 1-D tensor: `send` grouped by destination with scl[d] elements each, rcl[s] elements expected from
    rank s; in rounds of chunk_elems() per pair.  Returns the received tensor (grouped by source) on cdev."""
    import time
    t_begin = time.time()
    out = _all_to_all_v(send, scl, rcl, cdev, group)
    if cdev.type == "cuda" and coll_stats():
        torch.cuda.synchronize()
    me = dist.get_rank(group)
    es = send.element_size()
    note(name, es * (sum(scl) - scl[me]), es * (sum(rcl) - rcl[me]), time.time() - t_begin)
    return out


def _all_to_all_v(send, scl, rcl, cdev, group=None):
    world = dist.get_world_size(group)
    C = chunk_elems()
    n_in = sum(rcl)
    out = torch.empty(n_in, dtype=send.dtype, device=cdev)
    rounds = (max(list(scl) + list(rcl) + [0]) + C - 1) // C
    rt = torch.tensor([rounds], dtype=torch.int64, device=cdev)
    dist.all_reduce(rt, op=dist.ReduceOp.MAX, group=group)
    rounds = int(rt.item())
    if rounds <= 1:
        dist.all_to_all_single(out, send[:sum(scl)].to(cdev), list(rcl), list(scl), group=group)
        return out
    soff = np.concatenate([[0], np.cumsum(scl)]).astype(np.int64)
    roff = np.concatenate([[0], np.cumsum(rcl)]).astype(np.int64)
    for j in range(rounds):
        sj = [int(min(max(scl[d] - j * C, 0), C)) for d in range(world)]
        rj = [int(min(max(rcl[d] - j * C, 0), C)) for d in range(world)]
        pieces = [send[int(soff[d]) + j * C: int(soff[d]) + j * C + sj[d]] for d in range(world) if sj[d]]
        sb = (torch.cat(pieces) if pieces else send[:0]).to(cdev)
        rb = torch.empty(sum(rj), dtype=send.dtype, device=cdev)
        dist.all_to_all_single(rb, sb, rj, sj, group=group)
        at = 0
        for d in range(world):
            if rj[d]:
                out[int(roff[d]) + j * C: int(roff[d]) + j * C + rj[d]] = rb[at:at + rj[d]]
                at += rj[d]
    return out


def all_to_all_pairs(keys, counts, send_counts, group=None, name="bucket exchange (k1-mer keys + counts, all-to-all)"):
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
    if world == 1:                                  # nothing travels
        return keys[:scl[0]], counts[:scl[0]], rcl
    rk = all_to_all_v(keys, scl, rcl, cdev, group, name)
    rcn = all_to_all_v(counts, scl, rcl, cdev, group, name)
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


def all_to_all_bytes(bufs, device, group=None, name="read exchange (capped reads to partition owners, all-to-all)"):
    """bufs[d] = contiguous uint8 numpy array for rank d.  One all-to-all(v) of bytes (RCCL: staged through device
    memory; gloo: host memory).  Returns the arrays received, indexed by source rank."""
    world = dist.get_world_size(group)
    cdev = coll_device(device, group)
    sc = [int(b.size) for b in bufs]
    st = torch.tensor(sc, dtype=torch.int64, device=cdev)
    rt = torch.empty(world, dtype=torch.int64, device=cdev)
    dist.all_to_all_single(rt, st, group=group)
    rc = [int(v) for v in rt.cpu().tolist()]
    send = torch.from_numpy(np.concatenate(bufs) if sum(sc) else np.zeros(0, np.uint8))
    recv = all_to_all_v(send, sc, rc, cdev, group, name)
    out = recv.cpu().numpy()
    offs = np.concatenate([[0], np.cumsum(rc)]).astype(np.int64)
    return [out[offs[i]:offs[i + 1]] for i in range(world)]


def pack_read_pieces(items):
    """items: [(partition index, global doubled indices int64[n], (rows uint8[n, L], rc flags uint8[n]))] -> one uint8 array."""
    head = [len(items)]
    body = []
    for p, gidx, (rows, rc1) in items:
        n, L = int(rows.shape[0]), int(rows.shape[1]) if rows.ndim == 2 else 0
        head += [int(p), n, L]
        body += [np.ascontiguousarray(gidx, dtype=np.int64).view(np.uint8).reshape(-1), np.ascontiguousarray(rows, dtype=np.uint8).reshape(-1),
                 np.ascontiguousarray(rc1, dtype=np.uint8).reshape(-1)]
    return np.concatenate([np.asarray(head, dtype=np.int64).view(np.uint8)] + body)


def unpack_read_pieces(buf):
    """inverse of pack_read_pieces (views into buf)."""
    if buf.size == 0:
        return []
    n_items = int(buf[:8].view(np.int64)[0])
    head = buf[8:8 + 24 * n_items].view(np.int64).reshape(n_items, 3)
    pos = 8 + 24 * n_items
    out = []
    for p, n, L in head.tolist():
        gidx = buf[pos:pos + 8 * n].view(np.int64); pos += 8 * n
        rows = buf[pos:pos + n * L].reshape(n, L); pos += n * L
        rc1 = buf[pos:pos + n]; pos += n
        out.append((p, gidx, (rows, rc1)))
    return out
