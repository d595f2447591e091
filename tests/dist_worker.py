"""Worker for tests/test_distributed_cpu.py: runs shannon_amd.distributed.assemble_distributed with
world_size>1 on the gloo backend, per-rank compute supplied by the oracle (no GPU here)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist
from golden_util import MANIFEST, load_inputs, load_case, part_vectors
from oracle import seqs, count, extension, partition, sparse_flow as osf
from shannon_amd import distributed, exchange, kmers_for_component as kfc


class OracleOps(object):
    def __init__(self, r1, r2, K):
        self.r1, self.r2, self.K = r1, r2, K
        self.paired = r2 is not None
        self.device = torch.device("cpu")
        n = len(r1)
        self.store = kfc.ReadStore(r1, r2)

    supports_strand_specific = True
    strand_specific = False              # set by assemble_distributed
    min_weight, min_length, kmer_hard_cutoff = 3, 75, 1      # set by assemble_distributed

    def n_reads(self):
        return len(self.r1)

    owner_labelling = False              # True: the default path of GpuOps -- components labelled on owner shards (tests/cc_reference.py)

    def local_pairs(self, W, by_minimizer=False):
        k1 = self.K + 1
        self._owner = (lambda keys: exchange.owner_of_minimizer(keys, k1, not self.strand_specific, W)) if by_minimizer else (lambda keys: exchange.owner_of(keys, W))
        return self._local_pairs(W)

    def owned_table(self, rk, rc):
        k, c = self.reduce_pairs(rk, rc)
        return k.numpy().view(np.uint64), c.numpy()

    def component_table(self, owned, group, tick):
        """the steps of the owner-shard labelling in numpy (tests/cc_reference.py) behind the product's choreography"""
        from cc_reference import NumpyComponents
        W, rank = dist.get_world_size(group), dist.get_rank(group)
        cc = NumpyComponents(owned[0], owned[1], W, rank, self.K + 1, not self.strand_specific)
        (rk, rc), n_glob = distributed.component_table(cc, group, tick, None, self.device)
        self.component_keys = (owned[0], rk.numpy().view(np.uint64).copy())
        return (rk, rc), n_glob

    def _local_pairs(self, W):
        k1 = self.K + 1
        if self.strand_specific:         # forward counting of reads_1 and RC(reads_2): every key as it is
            recs = [r for f in seqs.strand_specific(list(self.r1), list(self.r2) if self.paired else None) for r in f]
            ck, cc = count.count_k1mers_packed(recs, k1)
            cc = cc.astype(np.int64)
            own = self._owner(ck)
            order = np.argsort(own, kind="stable")
            per = np.bincount(own, minlength=W)
            return torch.as_tensor(ck[order].view(np.int64)), torch.as_tensor(cc[order].astype(np.int32)), per
        recs = list(self.r1) + (list(self.r2) if self.paired else [])
        recs = recs + [seqs.reverse_complement(r) for r in recs]
        keys, cnts = count.count_k1mers_packed(recs, k1)
        canon = np.array([int(k) <= count.rc_key(k, k1) for k in keys], dtype=bool)
        pal = np.array([int(k) == count.rc_key(k, k1) for k in keys], dtype=bool)
        ck, cc = keys[canon], cnts[canon].astype(np.int64)
        cc[pal[canon]] //= 2
        own = self._owner(ck)
        order = np.argsort(own, kind="stable")
        per = np.bincount(own, minlength=W)
        return torch.as_tensor(ck[order].view(np.int64)), torch.as_tensor(cc[order].astype(np.int32)), per

    def reduce_pairs(self, rk, rc):
        k = rk.numpy().view(np.uint64)
        uk, inv = np.unique(k, return_inverse=True)
        s = np.bincount(inv, weights=rc.numpy().astype(np.float64), minlength=len(uk)).astype(np.int32)
        if self.kmer_hard_cutoff > 1:        # `jellyfish dump -L` on the reduced counts; a canonical palindrome stands for 2 x its count
            k1 = self.K + 1
            pal = np.zeros(len(uk), dtype=bool) if self.strand_specific else np.array([int(k) == count.rc_key(k, k1) for k in uk], dtype=bool)
            keep = np.where(pal, 2 * s.astype(np.int64), s.astype(np.int64)) >= self.kmer_hard_cutoff
            uk, s = uk[keep], s[keep]
        return torch.as_tensor(uk.view(np.int64)), torch.as_tensor(s)

    def table_from_pairs(self, gk, gc):
        k1 = self.K + 1
        tab = {}
        if self.strand_specific:
            return {count.key_to_str(k, k1): c for k, c in zip(gk.numpy().view(np.uint64).tolist(), gc.numpy().tolist())}
        for k, c in zip(gk.numpy().view(np.uint64).tolist(), gc.numpy().tolist()):
            r = count.rc_key(k, k1)
            if r == k:
                tab[count.key_to_str(k, k1)] = 2 * c
            else:
                tab[count.key_to_str(k, k1)] = c
                tab[count.key_to_str(r, k1)] = c
        return tab

    def extension(self, tab, partition_size, group=None, presharded=None):
        if presharded is not None:
            # this rank's whole components: checked against the components of the job's k1-mers, then the replicated oracle extension
            # on the tables of all ranks together (the sharded walks and their merge are GpuOps' own: tests/test_distributed_gpu.py)
            from cc_reference import reference_labels
            rk, rc = tab
            parts = [None] * dist.get_world_size(group)
            dist.all_gather_object(parts, (rk.numpy().view(np.uint64), rc.numpy(), self.component_keys[0]), group=group)
            keys = np.concatenate([p[0] for p in parts])
            assert len(keys) == presharded == len(np.unique(keys)) == sum(len(p[2]) for p in parts)
            assert np.array_equal(np.sort(keys), np.sort(np.concatenate([p[2] for p in parts])))       # the owned shards, moved, not changed
            lab = reference_labels(keys, self.K + 1, not self.strand_specific)
            where = np.repeat(np.arange(len(parts)), [len(p[0]) for p in parts])
            assert len(np.unique(np.stack([lab, where], axis=1), axis=0)) == len(np.unique(lab))       # a component lives on ONE rank
            tab = self.table_from_pairs(torch.as_tensor(keys.view(np.int64)), torch.as_tensor(np.concatenate([p[1] for p in parts])))
        return extension.run_correction([(k, tab[k]) for k in sorted(tab, reverse=True)], min_weight=self.min_weight, min_length=self.min_length,
                                        comp_size_threshold=partition_size)

    def route(self, res, K, partition_size, pv):
        pv = pv or []
        nc, k2c = partition.build_partitions([b[0] for b in res.big_components], [p[0] for p in pv], [p[1] for p in pv] if pv else None,
                                             res.remaining, res.allowed, K)
        files, _ = partition.partition_k1mers(nc, k2c, K)
        routes = {n: [] for n in nc}
        N = len(self.r1)
        if self.strand_specific:         # plain read indices; the pair of read d is (reads_1[d], RC(reads_2[d]))
            for d in range(N):
                a = self.r1[d]
                if a.strip("ACTG"):
                    continue
                cs = partition.get_comps(a, k2c, K)
                if self.paired:
                    b = seqs.reverse_complement(self.r2[d])
                    if b.strip("ACTG"):
                        continue
                    cs = cs | partition.get_comps(b, k2c, K)
                for c in cs:
                    routes[c].append(d)
            return {"new_components": nc, "k1mers": files, "routes": {n: np.array(v, dtype=np.uint32) for n, v in routes.items()}}
        for d in range(2 * N):
            a = self.store.mate1(d)
            if a.strip("ACTG"):
                continue
            cs = partition.get_comps(a, k2c, K)
            if self.paired:
                b = self.store.mate2(d)
                if b.strip("ACTG"):
                    continue
                cs = cs | partition.get_comps(b, k2c, K)
            for c in cs:
                routes[c].append(d)
        return {"new_components": nc, "k1mers": files, "routes": {n: np.array(v, dtype=np.uint32) for n, v in routes.items()}}

    def collect(self, sel):
        if self.strand_specific:
            return [(self.r1[int(d)], seqs.reverse_complement(self.r2[int(d)]) if self.paired else None) for d in sel]
        return [(self.store.mate1(int(d)), self.store.mate2(int(d)) if self.paired else None) for d in sel]

    def n_nodes(self, part, name, K):
        from shannon_amd.pipeline import n_kmer_nodes
        return n_kmer_nodes(part["k1mers"][name], K)

    def graph(self, part, name, pieces, K, paired):
        from shannon_amd import mbgraph_native
        recs = sorted((int(g), m) for gidx, data in pieces for g, m in zip(gidx.tolist(), data))
        reads = [[m[0] for _, m in recs], [m[1] for _, m in recs]] if paired else [[m[0] for _, m in recs]]
        singles, comps, _log = mbgraph_native.run_partition(part["k1mers"][name], reads, K, paired)   # native host code, seed scans on the CPU
        return singles, comps

    def sparse_flow(self, flat, ids, seed):
        return [osf.sparse_flow_component(nd, ed, pt, seed=seed, comp_id=c) for (nd, ed, pt), c in zip(flat, ids)]


def main():
    name, out = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    rank, W = dist.get_rank(), dist.get_world_size()
    m = MANIFEST[name]
    g = load_case(name)
    inp = load_inputs(name)
    n = len(inp[0])
    lo, hi = rank * n // W, (rank + 1) * n // W
    r1 = inp[0][lo:hi]
    r2 = inp[1][lo:hi] if m["paired"] else None
    psize = m.get("partition_size", 500)
    pv = [part_vectors(len(b["contigs"]), psize) for b in g["big_components"]] or None
    ops = OracleOps(r1, r2, m["K"])
    ops.owner_labelling = os.environ.get("SHN_TEST_OWNER_LABELS") == "1"
    if len(sys.argv) > 3:                    # a rank whose graph stage fails: every rank must raise together (no rank left in the gather)
        fail = int(sys.argv[3])

        def graph_batch(*a, **k):
            if rank == fail:
                raise RuntimeError("boom on purpose")
            return None
        ops.graph_batch = graph_batch
        try:
            distributed.assemble_distributed(ops, m["K"], psize, "s", m["sf_seed"], pv)
            msg = "no error"
        except RuntimeError as ex:
            msg = str(ex)
        open(out + ".rank%d" % rank, "w").write(msg)
        dist.destroy_process_group()
        return
    res = distributed.assemble_distributed(ops, m["K"], psize, "s", m["sf_seed"], pv, double_stranded=not m.get("strand_specific"),
                                           min_weight=m.get("kmer_soft_cutoff", 3), kmer_hard_cutoff=m.get("kmer_hard_cutoff", 1))
    if rank == 0:
        json.dump({"partitions": dict(res["partitions"]), "final": res["final"], "contigs": res["contigs"],
                   "owner_labelling_ran": hasattr(ops, "component_keys"), "n_k1mers": res["n_k1mers"]}, open(out, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
