#!/usr/bin/env python3
"""Development aid: the contig stage both ways (pipelined host stage beside the walks / GPU rounds after them) on a synthetic
input, each twice -- contigs, connections and components must be identical in all four runs.

    python tools/check_contig_paths.py --genes 2000 --reads 10000000
"""
import argparse, hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genes", type=int, default=2000)
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--K", type=int, default=25)
    args = ap.parse_args()
    import bench
    from shannon_amd import device, extension_correction as ec
    r1, r2 = bench.gen_reads(args.reads // 2, 20240501, args.genes, torch.device("cuda", 0))
    ctx = device.Context(0)
    sets = [device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)]
    table = device.count_k1mers(ctx, sets, args.K + 1, True)
    print("k1-mers", len(table), flush=True)
    sigs = []
    for mode in ("0", "1", "0", "1"):
        os.environ["SHN_CONTIG_GPU"] = mode
        T = {}
        t0 = time.time()
        res = ec.run_correction(ctx, table, 3, 75, 500, want_allowed=False, timings=T)
        h = hashlib.sha256()
        h.update("\n".join(res.contigs).encode())
        hc = hashlib.sha256(np.asarray(res.conn_off, np.int64).tobytes() + np.asarray(res.conn_nb, np.int64).tobytes()
                            + np.asarray(res.conn_w, np.int64).tobytes()).hexdigest()[:12]
        hm = hashlib.sha256(np.asarray(res.comp_members, np.int64).tobytes()).hexdigest()[:12]
        sigs.append((h.hexdigest()[:12], hc, hm))
        print("SHN_CONTIG_GPU=%s: %.2f s, %d contigs, %d single, %d remaining bins, contigs %s connections %s components %s  %s"
              % (mode, time.time() - t0, len(res.contigs), len(res.single_contigs), len(res.remaining), sigs[-1][0], hc, hm,
                 {k: round(v, 3) for k, v in T.items()}), flush=True)
    print("ALL EQUAL" if len(set(sigs)) == 1 else "DIFFERENT", flush=True)


if __name__ == "__main__":
    main()
