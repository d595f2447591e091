#!/bin/bash
# A/B of the software-pipelined records kernel (round 5): the adjacency timer of bench.py under both builds of the same ABI
mkdir -p gpurun_out/ab
python -m pytest tests/test_extension_gpu.py -x -q -m gpu > gpurun_out/ab/ext_tests.txt 2>&1
for i in 1 2; do
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline --overlap-steps 0 > gpurun_out/ab/pipe_$i.json 2> gpurun_out/ab/pipe_$i.err
  SHN_HIP_LIB=$PWD/ab/libshannon_hip_nopipe.so python bench.py --steps 4 --warmup 1 --no-cpu-baseline --overlap-steps 0 > gpurun_out/ab/nopipe_$i.json 2> gpurun_out/ab/nopipe_$i.err
done
tail -3 gpurun_out/ab/ext_tests.txt
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        st=j.get("stages_ms") or j.get("stage_ms") or {}
        print(f, j["value"], j["ms_per_step"], {k:v for k,v in st.items() if "adjac" in k or "walk" in k}, j.get("digest"))
    except Exception as e: print(f, "ERR", e)
P
