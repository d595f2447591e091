cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
SHN_DEBUG=1 python bench.py --no-cpu-baseline --steps 1 --warmup 0 2>&1 >/dev/null | grep "converged\|memo_follow"
SHN_DEBUG=1 python bench.py --no-cpu-baseline --steps 1 --warmup 0 2>&1 >/dev/null | grep "round 6[0-9]" | tail -2 | cut -c1-140
python bench.py --no-cpu-baseline --reads 2000000 --genes 300 --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
h = d['config']['host_stage_seconds_per_step']
print('300 genes', round(d['ms_per_step'],1), d['config']['transcripts'], {k: round(v, 3) for k, v in h.items()})"
python bench.py --no-cpu-baseline --reads 80000000 --families 8 --steps 1 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
h = d['config']['host_stage_seconds_per_step']
print('8 families', round(d['ms_per_step'],1), d['config']['transcripts'], {k: round(v, 3) for k, v in h.items()})"
