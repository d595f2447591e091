#!/usr/bin/env python3
"""Development aid: the walks of one input under different block schedules (SHN_EXT_LIMIT0 / SHN_EXT_GROW / SHN_EXT_TAIL):
rounds, steps and time.  python tools/ext_blocks_probe.py --genes 20000 --reads 100000000"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genes", type=int, default=20000)
    ap.add_argument("--reads", type=int, default=100_000_000)
    ap.add_argument("--K", type=int, default=25)
    args = ap.parse_args()
    import bench
    from shannon_amd import device, extension_correction as ec
    dev = torch.device("cuda", 0)
    r1, r2 = bench.gen_reads(args.reads // 2, 20240501, args.genes, dev, read_seed=20240503)
    ctx = device.Context(0)
    sets = [device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)]
    table = device.count_k1mers(ctx, sets, args.K + 1, True)
    ns_guess = None
    configs = [("default", {}), ("default again", {})]
    for l0 in (8, 4, 2, 1):
        for tail in ("8", "0"):
            configs.append(("limit0 = ns/%d tail %s" % (l0, tail), {"DIV": l0, "SHN_EXT_TAIL": tail}))
    configs.append(("limit0 = ns/32 grow 16", {"DIV": 32, "SHN_EXT_GROW": "16"}))
    configs.append(("limit0 = ns/32 grow 8 tail 0", {"DIV": 32, "SHN_EXT_GROW": "8", "SHN_EXT_TAIL": "0"}))
    for name, env in configs:
        for k in ("SHN_EXT_LIMIT0", "SHN_EXT_GROW", "SHN_EXT_TAIL"):
            os.environ.pop(k, None)
        for k, v in env.items():
            if k == "DIV":
                if ns_guess:
                    os.environ["SHN_EXT_LIMIT0"] = str(max(4096, ns_guess // v))
            else:
                os.environ[k] = v
        ctx.timer_reset()
        t0 = time.time()
        ext = ec.Extension(ctx, table, 3)
        ctx.sync()
        dt = time.time() - t0
        ns_guess = ext.n_walks
        tm = ctx.timers()
        print("%-32s %.3f s  rounds %3d  steps %d  walk %.0f ms (thread %.0f, wave %.0f, mark %.0f) prepare %.0f" % (
            name, dt, ext.iterations, ext.total_steps, tm.get("extend.walk", (0, 0))[0], tm.get("extend.walk_thread", (0, 0))[0],
            tm.get("extend.walk_wave", (0, 0))[0], tm.get("extend.mark", (0, 0))[0], tm.get("extend.prepare", (0, 0))[0]), flush=True)
        ext.close()


if __name__ == "__main__":
    main()
