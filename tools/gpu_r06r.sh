#!/bin/bash
# Round 6, final session on the final library: the whole GPU suite, the regime table, the driver's line, config 3's counters
# (profiles/r06/gpu_tests.log, regimes.json, bench_line.json; profiles/r06_c3_bf16).     gpurun --timeout 3400 -- 'bash tools/gpu_r06r.sh'
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
( time timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -130 ) > gpurun_out/r06/gpu_tests_final.log 2>&1
tail -6 gpurun_out/r06/gpu_tests_final.log | cut -c1-200
PROFILE_STEPS=3 PROFILE_WARMUP=1 timeout 900 bash tools/profile_bench.sh r06r_c3 --residual --mlp-split-bf16 > gpurun_out/r06/profile_c3.log 2>&1
timeout 2400 python3 tools/regimes.py --tag r06 > gpurun_out/r06/regimes.log 2>&1
grep "two-launch\|headline\|C5\|split-bf16\|batch of 16" gpurun_out/r06/regimes.log | cut -c1-160
( time timeout 1500 python3 bench.py ) > gpurun_out/r06/bench_default_final.log 2>&1
tail -c 300 gpurun_out/r06/bench_default_final.log
