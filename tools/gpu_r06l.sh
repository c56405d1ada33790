#!/bin/bash
# Round 6, final session: the whole GPU suite on the final library, the regime table and the driver's line
# (profiles/r06/gpu_tests.log, regimes.json, bench_line.json).     gpurun --timeout 3400 -- 'bash tools/gpu_r06l.sh'
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
( time timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -130 ) > gpurun_out/r06/gpu_tests_final.log 2>&1
tail -6 gpurun_out/r06/gpu_tests_final.log | cut -c1-200
timeout 2400 python3 tools/regimes.py --tag r06 > gpurun_out/r06/regimes.log 2>&1
grep "two-launch\|headline\|C5\|split-bf16\|batch of 16" gpurun_out/r06/regimes.log | cut -c1-160
( time timeout 1500 python3 bench.py ) > gpurun_out/r06/bench_default_final.log 2>&1
tail -c 300 gpurun_out/r06/bench_default_final.log
