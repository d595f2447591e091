"""Repeat one shared-GPU N-rank case many times and compare every run's result with the first (round 5: a strand-specific world-2 run
gave other contigs once).  usage: python tools/stress_dist_r05.py <repeats> <world> <paired 0/1> <genes> <seed> <ss 0/1> [env K=V ...]"""
import json, os, subprocess, sys, hashlib, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rep, world, paired, genes, seed, ss = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6] == "1"
env = dict(os.environ, MASTER_ADDR="127.0.0.1")
for kv in sys.argv[7:]:
    k, v = kv.split("=", 1)
    env[k] = v
first = None
bad = 0
t0 = time.time()
for i in range(rep):
    out = "/tmp/stress_dist_%d.json" % os.getpid()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", str(29700 + (i % 50)), os.path.join(ROOT, "tests", "dist_gpu_worker.py"), paired, genes, seed, "12000", out]
                       + (["ss"] if ss else []), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
    if p.returncode != 0:
        print("run", i, "FAILED rc", p.returncode, p.stdout[-1500:])
        bad += 1
        continue
    got = json.load(open(out))
    key = hashlib.sha256(json.dumps([got["contigs"], got["final"]], sort_keys=True).encode()).hexdigest()[:16]
    if first is None:
        first = (key, got)
    elif key != first[0]:
        bad += 1
        a, b = set(first[1]["contigs"]), set(got["contigs"])
        print("run", i, "DIFFERS: contigs", len(first[1]["contigs"]), "->", len(got["contigs"]), "only in first", len(a - b), "only here", len(b - a),
              "same order otherwise" if [c for c in got["contigs"] if c in a] == [c for c in first[1]["contigs"] if c in b] else "order differs",
              "table sizes", got.get("table_sizes"), "first", first[1].get("table_sizes"))
        print("   digests first", first[1].get("digests"))
        print("   digests here ", got.get("digests"))
        for c in sorted(a - b, key=len)[:3]:
            print("   gone :", len(c), c[:60])
        for c in sorted(b - a, key=len)[:3]:
            print("   new  :", len(c), c[:60])
print("%d runs, %d differ / fail, %.0f s, env %s" % (rep, bad, time.time() - t0, sys.argv[7:]))
