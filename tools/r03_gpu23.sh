#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03v; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests/test_e2e_gpu.py tests/test_post_gpu.py tests/test_lp_gpu.py tests/test_reference_api_gpu.py tests/test_random_parity_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -30 $O/tests.log | cut -c1-300
