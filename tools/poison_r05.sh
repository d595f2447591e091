#!/bin/bash
# round 5: where does a poisoned (never-written) device buffer get read?  python -X faulthandler names the host frame of a crash.
mkdir -p gpurun_out/poison
leg() { name=$1; shift; echo "=== $name: $*"; ( timeout 300 env "$@" > gpurun_out/poison/$name.log 2>&1; echo "rc=$?" >> gpurun_out/poison/$name.log ); tail -25 gpurun_out/poison/$name.log; }
P="python -X faulthandler tools/stress_digest.py --repeats 6"
leg a5_onerun      SHN_DEV_POISON=165 $P --pipeline none
leg a5_onerun_nows SHN_DEV_POISON=165 SHN_DEV_POISON_WS=0 $P --pipeline none
leg a5_pipe0       SHN_DEV_POISON=165 $P --pipeline 0
leg a5_pipe1       SHN_DEV_POISON=165 $P --pipeline 1
leg a5_assemble    SHN_DEV_POISON=165 $P --pipeline none --assemble-every 1
leg a5_assemble_nows SHN_DEV_POISON=165 SHN_DEV_POISON_WS=0 $P --pipeline none --assemble-every 1
leg ff_assemble    SHN_DEV_POISON=255 $P --pipeline both --assemble-every 1
