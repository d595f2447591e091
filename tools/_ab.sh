cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1f
python -m pytest tests/test_extension_gpu.py tests/test_e2e_gpu.py -x -q -m gpu 2>&1 | tail -3
SHN_DEBUG=1 python bench.py --no-cpu-baseline --steps 1 --warmup 0 > gpurun_out/r1f/dbg.json 2> gpurun_out/r1f/dbg.err
for v in 1 2; do
  python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_ms_per_step']
print(round(d['ms_per_step'],1), d['config']['extension_iterations'], d['config']['transcripts'], 'thread', round(k['extend.walk_thread'],1), 'wave', round(k['extend.walk_wave'],1), 'mark', round(k['extend.mark'],1), 'walk', round(k['extend.walk'],1))"
done
