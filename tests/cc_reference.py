"""The k1-mer graph's components and the steps of the owner-shard labelling restated in numpy + scipy -- TEST INFRASTRUCTURE.
  reference_labels     components of a set of k1-mers by the rule of the labelling kernels (extension_correction.py:202-245, 372-390:
                       adjacent k1-mers, and k1-mers that share a K-mer at the same end; low-complexity k1-mers stand alone)
  NumpyComponents      the backend of shannon_amd.distributed.component_table (what GpuOps does with shn_cc_*) on host arrays: the CPU
                       tests run the N-rank choreography of the default path over gloo with it"""
import numpy as np
import torch


def rc_keys(keys, k):
    out = np.zeros_like(keys)
    x = keys.copy()
    for _ in range(k):
        out = (out << np.uint64(2)) | (np.uint64(3) - (x & np.uint64(3)))
        x >>= np.uint64(2)
    return out


def low_complexity(keys, k):
    cnt = np.zeros((4, len(keys)), dtype=np.int64)
    x = keys.copy()
    for _ in range(k):
        b = (x & np.uint64(3)).astype(np.int64)
        for v in range(4):
            cnt[v] += b == v
        x >>= np.uint64(2)
    return cnt.max(axis=0) >= k - 2


def which_keys(sk, k, canonical):
    """for the sorted keys sk: 16 x (neighbour / sibling key, is it one) in the order of the kernels (8 neighbours, 8 siblings)"""
    mask = np.uint64((1 << (2 * k)) - 1) if k < 32 else np.uint64(0xFFFFFFFFFFFFFFFF)
    sh = np.uint64(2 * (k - 1))
    for which in range(16):
        b = np.uint64(which & 3)
        ok = np.ones(len(sk), dtype=bool)
        if which < 4:
            y = ((sk << np.uint64(2)) | b) & mask
        elif which < 8:
            y = (sk >> np.uint64(2)) | (b << sh)
        elif which < 12:                                             # siblings: the same K-prefix, another last base
            y = (sk & ~np.uint64(3)) | b
            ok = (sk & np.uint64(3)) != b
        else:                                                        # ... the same K-suffix, another first base
            y = (sk & ~(np.uint64(3) << sh)) | (b << sh)
            ok = ((sk >> sh) & np.uint64(3)) != b
        if canonical:
            y = np.minimum(y, rc_keys(y, k))
        yield y, ok


def _edges_inside(sk, alive, k, canonical):
    src, dst = [], []
    for y, ok in which_keys(sk, k, canonical):
        pos = np.minimum(np.searchsorted(sk, y), max(len(sk) - 1, 0))
        hit = ok & alive & (sk[pos] == y) & alive[pos] if len(sk) else np.zeros(0, bool)
        src.append(np.nonzero(hit)[0])
        dst.append(pos[hit])
    return np.concatenate(src), np.concatenate(dst)


def _components(n, src, dst):
    """label = smallest member of the component"""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    _, lab = connected_components(coo_matrix((np.ones(len(src), dtype=np.int8), (src, dst)), shape=(n, n)), directed=False)
    first = np.full(lab.max() + 1, n, dtype=np.int64)
    np.minimum.at(first, lab, np.arange(n, dtype=np.int64))
    return first[lab]


def reference_labels(keys, k, canonical):
    """component number of every key (keys: distinct, any order)"""
    order = np.argsort(keys)
    sk = keys[order]
    src, dst = _edges_inside(sk, ~low_complexity(sk, k), k, canonical)
    out = np.empty(len(sk), dtype=np.int64)
    out[order] = _components(len(sk), src, dst)
    return out


def same_partition(a, b):
    pairs = np.unique(np.stack([a, b], axis=1), axis=0)
    return len(pairs) == len(np.unique(a)) == len(np.unique(b))


class NumpyComponents(object):
    """keys (uint64, distinct), counts of this rank's shard; world, rank; k1, canonical"""

    def __init__(self, keys, counts, world, rank, k, canonical):
        from shannon_amd import exchange
        self.exchange = exchange
        order = np.argsort(keys)
        self.keys, self.counts = np.asarray(keys, dtype=np.uint64)[order], np.asarray(counts)[order]
        self.W, self.rank, self.k, self.canonical = world, rank, k, canonical
        self.n = len(self.keys)
        self.alive = ~low_complexity(self.keys, k) if self.n else np.zeros(0, bool)
        src, dst = _edges_inside(self.keys, self.alive, k, canonical) if self.n else (np.zeros(0, np.int64), np.zeros(0, np.int64))
        self.lab = _components(self.n, src, dst)

    def queries(self):
        qk, ql, qd = [], [], []
        for y, ok in which_keys(self.keys, self.k, self.canonical):
            dest = self.exchange.owner_of_minimizer(y, self.k, self.canonical, self.W)
            want = ok & self.alive & (dest > self.rank)
            qk.append(y[want]); ql.append(self.lab[want]); qd.append(dest[want])
        qk, ql, qd = np.concatenate(qk), np.concatenate(ql), np.concatenate(qd)
        order = np.argsort(qd, kind="stable")
        per = np.bincount(qd, minlength=self.W).astype(np.int64)
        return torch.as_tensor(qk[order].view(np.int64)), torch.as_tensor(ql[order].astype(np.int32)), per

    def answer(self, rk, rl, rcl, base):
        k = rk.numpy().view(np.uint64)
        l = rl.numpy().astype(np.int64)
        src = np.repeat(np.arange(self.W), np.asarray(rcl, dtype=np.int64))
        if not self.n or not len(k):
            return torch.zeros(0, dtype=torch.int64)
        pos = np.minimum(np.searchsorted(self.keys, k), self.n - 1)
        hit = (self.keys[pos] == k) & self.alive[pos]
        e = np.stack([base[self.rank] + self.lab[pos[hit]], np.asarray(base, dtype=np.int64)[src[hit]] + l[hit]], axis=1)
        return torch.as_tensor(e.reshape(-1).astype(np.int64))

    def solve(self, ge, id_limit):
        e = ge.numpy().reshape(-1, 2)
        ids, inv = np.unique(e.reshape(-1), return_inverse=True)
        inv = inv.reshape(-1, 2)
        lab = _components(len(ids), inv[:, 0], inv[:, 1])
        assert not len(ids) or ids.max() < id_limit
        return torch.as_tensor(ids), torch.as_tensor(ids[lab] if len(ids) else ids)

    def labels(self, base_me, ids, labels):
        g = base_me + self.lab
        ids, labels = ids.numpy(), labels.numpy()
        if len(ids):
            pos = np.minimum(np.searchsorted(ids, g), len(ids) - 1)
            hit = ids[pos] == g
            g = np.where(hit, labels[pos], g)
        return torch.as_tensor(g.astype(np.int64))

    def sizes(self, glabel, at_least):
        u, c = np.unique(glabel.numpy(), return_counts=True)
        return u[c >= at_least], c[c >= at_least]

    def shard(self, glabel, big, big_owner):
        g = glabel.numpy()
        with np.errstate(over="ignore"):
            own = (self.exchange.fmix64_np(g.astype(np.uint64) ^ np.uint64(0x5851F42D4C957F2D)) % np.uint64(self.W)).astype(np.int64)
        if len(big):
            pos = np.minimum(np.searchsorted(big, g), len(big) - 1)
            hit = big[pos] == g
            own = np.where(hit, np.asarray(big_owner, dtype=np.int64)[pos], own)
        order = np.argsort(own, kind="stable")
        self.owner = own
        return (torch.as_tensor(self.keys[order].view(np.int64)), torch.as_tensor(self.counts[order].astype(np.int32)),
                np.bincount(own, minlength=self.W).astype(np.int64))

    def close(self):
        pass
