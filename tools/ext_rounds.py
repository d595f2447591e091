"""Per-round log of the extension on the benchmark input (SHN_DEBUG output of shn_extend), summarised per rank block."""
import os, sys, re, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "run":
    sys.path.insert(0, ROOT)
    import torch, bench
    from shannon_amd import device, extension_correction as ec
    dev = torch.device("cuda", 0)
    r1, r2 = bench.gen_reads(5_000_000, 20240501, 1, dev)
    ctx = device.Context(0)
    t = device.count_k1mers(ctx, [device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)], 26, True)
    e = ec.Extension(ctx, t, 3); e.close()          # warm
    os.environ["SHN_DEBUG"] = "1"
    e = ec.Extension(ctx, t, 3); e.close()
    sys.exit(0)
out = subprocess.run([sys.executable, __file__, "run"], stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True).stderr
rows = re.findall(r"round (\d+) \[(\d+),(\d+)\): dirty=(\d+) long=(\d+) short=(\d+) changed_kmers=(\d+).*?longest wavefront walk: (\d+) steps, most sequential: (\d+).*?([\d.]+) ms", out)
blocks = {}
for r in rows:
    blocks.setdefault((int(r[1]), int(r[2])), []).append(r)
for (a, b), rs in blocks.items():
    ms = [float(r[9]) for r in rs]
    print("block [%d,%d): %d rounds, %.1f ms; first rounds (dirty, ms): %s ... last: %s" % (a, b, len(rs), sum(ms), [(int(r[3]), float(r[9])) for r in rs[:6]], [(int(r[3]), float(r[9])) for r in rs[-4:]]))
    print("    longest walk / most sequential steps per round:", [(int(r[7]), int(r[8])) for r in rs[:12]])
