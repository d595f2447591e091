#!/bin/bash
# round 5, last measurements: the GPU suite, smoke, the default bench line, and the N-rank evidence (labelling timings, repeat-run
# stress of the labelling and of a 2-rank job on both paths, the serialized N-rank model) -> gpurun_out/final/ (copied to profiles/r05_*)
O=gpurun_out/final; mkdir -p $O
timeout 1300 python -m pytest tests -x -q -m gpu --durations=8 > $O/gpu_suite.txt 2>&1; tail -4 $O/gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 600 python bench.py > $O/bench_config2.json 2> $O/bench_config2.err; tail -c 600 $O/bench_config2.json | head -c 300; echo
timeout 300 python tools/cc_timing.py > $O/cc_timing.txt 2>&1; tail -14 $O/cc_timing.txt | head -13
timeout 300 python tools/stress_cc_r05.py 300 2 1 strand-specific > $O/stress_cc.txt 2>&1; timeout 300 python tools/stress_cc_r05.py 300 3 0 canonical >> $O/stress_cc.txt 2>&1; grep "runs," $O/stress_cc.txt
timeout 400 python tools/stress_dist_r05.py 40 2 1 12 4 1 SHN_DIST_DIGEST=1 > $O/stress_dist.txt 2>&1; timeout 400 python tools/stress_dist_r05.py 40 2 1 12 4 1 SHN_OWNER_LABELS=0 >> $O/stress_dist.txt 2>&1; grep "runs," $O/stress_dist.txt
bash tools/serial_model_r05.sh > $O/serial_model.txt 2>&1; grep "value" $O/serial_model.txt
