cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1f
python -m pytest tests/test_extension_gpu.py tests/test_e2e_gpu.py -x -q -m gpu 2>&1 | tail -3
for v in 1 2 3; do
  timeout 120 python bench.py --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_ms_per_step']
print(round(d['ms_per_step'],1), d['config']['extension_iterations'], d['config']['transcripts'], 'ext', round(k['extend'],1), 'walk', round(k['extend.walk'],1), 'gpu_walks', round(d['config']['host_stage_seconds_per_step']['ext.gpu_walks'], 4))"
done
