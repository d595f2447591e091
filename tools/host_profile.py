"""Where the host time of a step goes (main thread, cProfile) at a bench workload.  usage: host_profile.py [genes] [reads] [K]
(defaults: BASELINE configs[2]); SHN_PARTS_TIMELINE=1 adds the per-partition time line of the graph stage."""
import os, sys, time, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench, shannon_amd
shannon_amd.malloc_tune()
from shannon_amd import device, pipeline, kmers_for_component as kfc
genes = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 25
dev = torch.device("cuda", 0)
r1, r2 = bench.gen_reads(reads // 2, 20240501, genes, dev)
ctx = device.Context(0)
sets = [device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)]
store = kfc.ReadStore(r1, r2)
T = {}
for _ in range(2):
    pipeline.assemble_resident(ctx, sets[0], sets[1], store, K=K, sample="bench", seed=1, timings=T, keep_partitioning=True)
T.clear()
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
os.environ["SHN_GRAPH_LAPS"] = os.environ.get("LAPS_MIN", "3000000")
R = pipeline.assemble_resident(ctx, sets[0], sets[1], store, K=K, sample="bench", seed=1, timings=T, keep_partitioning=True)
pr.disable()
print("step %.3f s, %d transcripts" % (time.time() - t0, len(R.final)))
for k, v in T.items():
    print("  %-28s %.3f" % (k, v))
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])
