for v in "" "SHN_EXT_REFILL=1" "SHN_EXT_PASSES=256,4096" "SHN_EXT_PASSES=32,512,8192"; do
  echo "== $v"
  env $v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --overlap-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernel_ms_per_step']; c=d['config']
print(round(d['ms_per_step']), c['transcripts_sha256_16'], 'walks', c['extension_walks'], 'steps', c['extension_walk_steps'], 'iters', c['extension_iterations'], {x:round(k[x],1) for x in k if x.startswith('extend')})
"
done
