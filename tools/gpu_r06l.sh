#!/bin/bash
# Round 6, final session: the whole GPU suite on the final library, the config-3 split-bf16 profile (traffic after the predicated seed stores), the driver's line.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
( time timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -120 ) > gpurun_out/r06/gpu_tests_final.log 2>&1
tail -6 gpurun_out/r06/gpu_tests_final.log | cut -c1-200
PROFILE_STEPS=3 PROFILE_WARMUP=1 timeout 1500 bash tools/profile_bench.sh r06_c3_bf16 --residual --mlp-split-bf16 > gpurun_out/r06/profile_c3.log 2>&1
python3 tools/summarize_profile.py gpurun_out/prof_r06_c3_bf16 gpurun_out/r06/profile_summary_c3_bf16 > gpurun_out/r06/summarize_c3.log 2>&1
( time timeout 1500 python3 bench.py ) > gpurun_out/r06/bench_default_final.log 2>&1
tail -c 1200 gpurun_out/r06/bench_default_final.log
