#!/bin/bash
# Round 6, final library: counters of config 3 with the exact-fp32 MLP and of config 5 (profiles/r06_c3, r06_c5), then the driver's line.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
PROFILE_STEPS=3 PROFILE_WARMUP=1 timeout 900 bash tools/profile_bench.sh r06s_c3 --residual > gpurun_out/r06/profile_c3_fp32.log 2>&1
PROFILE_STEPS=3 PROFILE_WARMUP=1 timeout 900 bash tools/profile_bench.sh r06s_c5 --views 17 --height 752 --width 2816 --instances 64 --samples 128 > gpurun_out/r06/profile_c5.log 2>&1
( time timeout 1500 python3 bench.py ) > gpurun_out/r06/bench_default_final.log 2>&1
tail -c 300 gpurun_out/r06/bench_default_final.log
