cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1f
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for v in 0 20 0 20; do
  SHN_EXT_COARSE=$v timeout 120 python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_ms_per_step']
print($v, round(d['ms_per_step'],1), d['config']['extension_iterations'], d['config']['transcripts'], 'thread', round(k['extend.walk_thread'],1), 'wave', round(k['extend.walk_wave'],1), 'mark', round(k['extend.mark'],1), 'walk', round(k['extend.walk'],1), 'ext', round(k['extend'],1), d['config']['host_stage_seconds_per_step'])"
done
