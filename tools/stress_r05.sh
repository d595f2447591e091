#!/bin/bash
# round 5: localise the run-to-run difference of count -> extension (GPUTEST_r04).  Each leg is time-bounded.
mkdir -p gpurun_out/stress
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
leg() { name=$1; shift; echo "=== $name: $*" ; ( timeout 900 env "$@" python tools/stress_digest.py --repeats 100000 --assemble-every 4 --seconds ${SECS:-240} > gpurun_out/stress/$name.log 2>&1 ; echo "rc=$?" >> gpurun_out/stress/$name.log ) ; tail -4 gpurun_out/stress/$name.log; }
leg legacy SHN_DEV_LEGACY=1
leg events SHN_X=1
leg poisonA5 SHN_DEV_POISON=165
leg poison00 SHN_DEV_POISON=0
leg serialize SHN_DEV_LEGACY=1 AMD_SERIALIZE_KERNEL=3
