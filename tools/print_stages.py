"""Print the per-stage seconds of a bench.py JSON line (last line of the given file)."""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
c = d["config"]
print("n_gpus %d  %.1f ms/step  %.1f M reads/s" % (d["n_gpus"], d["ms_per_step"], d["value"] / 1e6))
mx = c.get("host_stage_seconds_per_step_slowest_rank") or {}
r0 = c["host_stage_seconds_per_step"]
for k in (mx or r0):
    print("  %-22s slowest rank %7.1f ms   rank 0 %7.1f ms" % (k, mx.get(k, 0) * 1e3, r0.get(k, 0) * 1e3))
comp = sum(v for k, v in (mx or r0).items() if not k.startswith("x:"))
print("  compute stages (slowest rank each): %.1f ms;  transcripts %s, partitions %s, distinct k1-mers %s" % (comp * 1e3, c["transcripts"], c["partitions"], c["distinct_k1mers"]))
