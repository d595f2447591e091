#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03w; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_extension_gpu.py tests/test_routing_gpu.py tests/test_midsize_gpu.py tests/test_host_utils.py -m gpu -x -q > $O/tests_sort.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests_sort.log | cut -c1-200
timeout 900 python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 --overlap-steps 0 > $O/bench_sort.json 2> $O/bench_sort.err; echo "bench rc=$?"
python3 - $O/bench_sort.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c=d['config']
print(d['ms_per_step'], c.get('transcripts'), c.get('transcripts_sha256_16'))
k=d['kernel_ms_per_step']
print({n: round(k[n],1) for n in ('extend','extend.sort','contig.stage','route','extend.walk','count.total') if n in k})
print({kk: round(v,3) for kk,v in c['host_stage_seconds_per_step'].items() if not kk.startswith('route.')})
PY
