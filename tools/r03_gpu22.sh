#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03u; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_extension_gpu.py -m gpu -x -q > $O/tests2.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests2.log
timeout 900 python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 --overlap-steps 0 > $O/bench2.json 2> $O/bench2.err; echo "bench rc=$?"
python3 - $O/bench2.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config'].get('transcripts'), d['config'].get('transcripts_sha256_16'), d['config'].get('extension_iterations'))
k=d['kernel_ms_per_step']
for n in sorted(k,key=lambda n:-k[n])[:12]: print(n, round(k[n],1))
PY
