cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1f
python bench.py --no-cpu-baseline --reads 2000000 --genes 300 --steps 2 --warmup 1 2>gpurun_out/r1f/g300.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('300 genes', round(d['ms_per_step'],1), d['config']['transcripts'], d['config']['host_stage_seconds_per_step'])"
grep "mbgraph\] load" gpurun_out/r1f/g300.err | tail -4
