#!/bin/bash
# A/B of the records kernel (round 5): LDS look-ups (2 / 4 dictionary look-ups in flight) against the dictionary-only kernel
mkdir -p gpurun_out/ab
timeout 900 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "minimizer_bucketed" > gpurun_out/ab/e2e.txt 2>&1; tail -2 gpurun_out/ab/e2e.txt
timeout 900 python -m pytest tests/test_midsize_gpu.py -x -q -m gpu > gpurun_out/ab/mid.txt 2>&1; tail -2 gpurun_out/ab/mid.txt
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --overlap-steps 0"
$B > gpurun_out/ab/lds2.json 2> gpurun_out/ab/lds2.err
SHN_REC_LDS=0 $B > gpurun_out/ab/dict.json 2> gpurun_out/ab/dict.err
SHN_HIP_LIB=$PWD/ab/libshannon_hip_fly4.so $B > gpurun_out/ab/lds4.json 2> gpurun_out/ab/lds4.err
$B > gpurun_out/ab/lds2b.json 2> gpurun_out/ab/lds2b.err
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(j["value"]/1e6,2), round(j["ms_per_step"]), "adjacency", round(j["kernel_ms_per_step"]["extend.adjacency"],1), j["config"].get("transcripts_sha256_16"))
    except Exception as e: print(f, "ERR", e)
P
