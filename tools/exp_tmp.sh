mkdir -p gpurun_out/r02
for cfg in "T16:SHN_GRAPH_THREADS=16" "T32:SHN_GRAPH_THREADS=32" "T64m:MALLOC_TOP_PAD_=1073741824 MALLOC_TRIM_THRESHOLD_=17179869184 MALLOC_MMAP_THRESHOLD_=33554432"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  env $envs SHN_DEBUG=1 SHN_DEBUG_PARTS=1 timeout 600 python bench.py --steps 2 --warmup 1 > gpurun_out/r02/exp_$tag.json 2> gpurun_out/r02/exp_$tag.err
  echo "== $tag"; python - <<PY
import json
d=json.loads(open('gpurun_out/r02/exp_$tag.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], {k:round(v,2) for k,v in d['config']['host_stage_seconds_per_step'].items() if k in ('graph','sparse flow','post','extension','count')})
PY
  grep "reads=3625640" gpurun_out/r02/exp_$tag.err | grep "bridge_all\|load reads\|find_bridging\|known" | tail -8
  grep "stage wall" gpurun_out/r02/exp_$tag.err | tail -2
done
