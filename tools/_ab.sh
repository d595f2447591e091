cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1f
for cfg in "32 16 64" "32 4 64" "16 2 32" "32 16 64" "32 4 64" "16 2 32"; do
  set -- $cfg
  SHN_EXT_LONG_WALK=$1 SHN_EXT_MEMO_MIN=$2 SHN_EXT_PROMOTE=$3 timeout 120 python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_ms_per_step']
print('$cfg', round(d['ms_per_step'],1), d['config']['extension_iterations'], d['config']['transcripts'], 'ext', round(k['extend'],1), 'thread', round(k['extend.walk_thread'],1), 'wave', round(k['extend.walk_wave'],1), 'mark', round(k['extend.mark'],1), 'walk', round(k['extend.walk'],1))"
done
