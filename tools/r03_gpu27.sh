#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03z; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_extension_gpu.py tests/test_midsize_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log | cut -c1-200
for CFG in 4s 2; do
timeout 900 python3 bench.py --config $CFG --no-cpu-baseline --steps 3 --warmup 1 --overlap-steps 0 > $O/bench_$CFG.json 2> $O/bench_$CFG.err; echo "bench $CFG rc=$?"
python3 - $O/bench_$CFG.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c=d['config']
print(d['ms_per_step'], c.get('transcripts'), c.get('transcripts_sha256_16'))
print({k: round(v,3) for k,v in c['host_stage_seconds_per_step'].items() if not k.startswith('route.') and not k.startswith('ext.')})
k=d['kernel_ms_per_step']
print({n: round(k[n],1) for n in ('extend','extend.prepare','extend.adjacency','extend.walk','count.total') if n in k})
PY
done
