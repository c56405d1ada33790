#!/bin/bash
# Round 6, a full session (as run on the -DVSRD_CULL_INNER library before it became opt-in; works on any build): the whole GPU suite, counters of configs 2 and 5,
# the regime table, the driver's line.     gpurun --timeout 3000 -- 'bash tools/gpu_r06w.sh'
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
( time timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -140 ) > gpurun_out/r06/gpu_tests_final.log 2>&1
tail -6 gpurun_out/r06/gpu_tests_final.log | cut -c1-200
timeout 600 bash tools/profile_bench.sh r06w > gpurun_out/r06/profile_c2.log 2>&1
PROFILE_STEPS=3 PROFILE_WARMUP=1 timeout 600 bash tools/profile_bench.sh r06w_c5 --views 17 --height 752 --width 2816 --instances 64 --samples 128 > gpurun_out/r06/profile_c5.log 2>&1
timeout 1500 python3 tools/regimes.py --tag r06 > gpurun_out/r06/regimes.log 2>&1
grep "two-launch\|headline\|C5\|split-bf16\|batch of 16" gpurun_out/r06/regimes.log | cut -c1-160
( time timeout 900 python3 bench.py ) > gpurun_out/r06/bench_default_final.log 2>&1
tail -c 300 gpurun_out/r06/bench_default_final.log
