"""round 6: user + system CPU seconds of a bench run against its wall time (how busy the granted cpus are)
usage: python tools/cpu_time_r06.py <bench args...>"""
import resource, subprocess, sys, time, json
t = time.time()
p = subprocess.run([sys.executable, "bench.py"] + sys.argv[1:], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
wall = time.time() - t
ru = resource.getrusage(resource.RUSAGE_CHILDREN)
d = json.loads(p.stdout.strip().splitlines()[-1])
steps = d["steps"] + d["warmup"]
print("wall %.1f s, user %.1f s, system %.1f s -> %.1f cpus busy on average; %.0f ms per step, %d steps (+ start-up); voluntary / involuntary switches %d / %d"
      % (wall, ru.ru_utime, ru.ru_stime, (ru.ru_utime + ru.ru_stime) / wall, d["ms_per_step"], steps, ru.ru_nvcsw, ru.ru_nivcsw))
