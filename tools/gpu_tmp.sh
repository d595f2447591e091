#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03w; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests/test_e2e_gpu.py tests/test_random_parity_gpu.py tests/test_reference_api_gpu.py tests/test_seeds_gpu.py tests/test_distributed_gpu.py tests/test_stress_gpu.py -m gpu -x -q > $O/tests_br.log 2>&1; echo "tests rc=$?"; tail -1 $O/tests_br.log | cut -c1-200
PARITY_SS=mix timeout 900 python3 tools/random_parity.py 30 8100 > $O/parity_br.txt 2>&1; tail -1 $O/parity_br.txt
PARITY_SS=mix timeout 900 python3 tools/random_parity.py 8 8600 big > $O/parity_br2.txt 2>&1; tail -1 $O/parity_br2.txt
SHN_GRAPH_LAPS=3000000 timeout 900 python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 --overlap-steps 0 > $O/bench_br.json 2> $O/bench_br.err; echo "bench rc=$?"
grep -a "bridge_all  \|find_known_paths  \|kp (device)" $O/bench_br.err | tail -3
python3 - $O/bench_br.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c=d['config']
print(d['ms_per_step'], c.get('transcripts'), c.get('transcripts_sha256_16'))
print({kk: round(v,3) for kk,v in c['host_stage_seconds_per_step'].items() if not kk.startswith('route.') and not kk.startswith('ext.')})
PY
