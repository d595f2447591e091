cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1f
python -m pytest tests/test_e2e_gpu.py tests/test_routing_gpu.py tests/test_extension_gpu.py -x -q -m gpu 2>&1 | tail -3
SHN_DEBUG=1 python bench.py --no-cpu-baseline --reads 80000000 --families 8 --steps 1 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('8 families', round(d['ms_per_step'],1), d['config']['transcripts'], d['config']['host_stage_seconds_per_step'])"
python bench.py --no-cpu-baseline --reads 2000000 --genes 300 --steps 2 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('300 genes', round(d['ms_per_step'],1), d['config']['transcripts'], d['config']['host_stage_seconds_per_step'])"
