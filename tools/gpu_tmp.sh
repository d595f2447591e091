#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03w; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests/test_e2e_gpu.py tests/test_random_parity_gpu.py tests/test_reference_api_gpu.py tests/test_distributed_gpu.py tests/test_routing_gpu.py -m gpu -x -q > $O/tests_ft.log 2>&1; echo "tests rc=$?"; tail -1 $O/tests_ft.log | cut -c1-200
timeout 900 python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 --overlap-steps 0 > $O/bench_ft.json 2> $O/bench_ft.err; echo "bench rc=$?"
python3 - $O/bench_ft.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c=d['config']
print(d['ms_per_step'], c.get('transcripts'), c.get('transcripts_sha256_16'))
print({kk: round(v,3) for kk,v in c['host_stage_seconds_per_step'].items() if not kk.startswith('route.') and not kk.startswith('ext.')})
PY
