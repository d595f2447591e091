#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03y; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests/test_e2e_gpu.py tests/test_random_parity_gpu.py tests/test_seeds_gpu.py tests/test_reference_api_gpu.py tests/test_distributed_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log | cut -c1-200
timeout 900 python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 --overlap-steps 0 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c=d['config']
print(d['ms_per_step'], c.get('transcripts'), c.get('transcripts_sha256_16'))
print({k: round(v,3) for k,v in c['host_stage_seconds_per_step'].items() if not k.startswith('route.') and not k.startswith('ext.')})
PY
