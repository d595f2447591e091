#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03t; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_distributed_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 1200 python3 bench.py --no-cpu-baseline --force-distributed --steps 2 --warmup 1 --overlap-steps 0 > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "dist1 rc=$?"
tail -3 $O/bench_dist1.err
python3 - $O/bench_dist1.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c=d['config']
print(d['ms_per_step'], c.get('transcripts'), c.get('transcripts_sha256_16'), c.get('rccl_ranks'))
print(json.dumps(c.get('host_stage_seconds_per_step'))[:1500])
PY
