"""sum the [mbgraph] laps of a host_profile.py run (stderr) by phase"""
import re, sys, collections
tot = collections.defaultdict(float); cnt = collections.Counter()
for l in open(sys.argv[1]):
    if not l.startswith("[mbgraph]"):
        continue
    m = re.match(r"\[mbgraph\]\s+(.*?)\s+([\d.]+) s\b", l)
    if m:
        tot[m.group(1).strip()] += float(m.group(2)); cnt[m.group(1).strip()] += 1
    m = re.match(r"\[mbgraph\]\s+kp \(device\) node text ([\d.]+) s scan ([\d.]+) s slow index ([\d.]+) s search ([\d.]+) s", l)
    if m:
        for k, v in zip(("kp node text", "kp scan", "kp slow index", "kp search"), m.groups()):
            tot["  " + k] += float(v); cnt["  " + k] += 1
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print("%-34s %8.3f thread-s  %5d laps" % (k, v, cnt[k]))
