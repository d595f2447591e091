cd $GRAFT_REPO_ROOT
python -m pytest tests/test_e2e_gpu.py tests/test_extension_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1
SHN_DEBUG=1 python bench.py --no-cpu-baseline --reads 2000000 --genes 300 --steps 2 --warmup 1 2>gpurun_out/g300.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
h = d['config']['host_stage_seconds_per_step']
print('300 genes', round(d['ms_per_step'],1), d['config']['transcripts'], {k: round(v, 3) for k, v in h.items()})"
grep "contig_graph\] [0-9]" gpurun_out/g300.err | tail -1
SHN_DEBUG=1 python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>gpurun_out/c1.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
h = d['config']['host_stage_seconds_per_step']
print('config1', round(d['ms_per_step'],1), d['config']['transcripts'], {k: round(v, 3) for k, v in h.items()})"
grep "contig_graph\] [0-9]" gpurun_out/c1.err | tail -1
