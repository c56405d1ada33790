#!/bin/bash
# Round 6, final session: the whole GPU suite on the final library and the driver's line (profiles/r06/gpu_tests.log, bench_line.json).
#   gpurun --timeout 3000 -- 'bash tools/gpu_r06l.sh'     [C3=1: also the config-3 split-bf16 profile, profiles/r06_c3_bf16]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
( time timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -120 ) > gpurun_out/r06/gpu_tests_final.log 2>&1
tail -6 gpurun_out/r06/gpu_tests_final.log | cut -c1-200
if [ "${C3:-0}" = 1 ]; then
  PROFILE_STEPS=3 PROFILE_WARMUP=1 timeout 1500 bash tools/profile_bench.sh r06_c3_bf16 --residual --mlp-split-bf16 > gpurun_out/r06/profile_c3.log 2>&1
  python3 tools/summarize_profile.py gpurun_out/prof_r06_c3_bf16 gpurun_out/r06/profile_summary_c3_bf16 > gpurun_out/r06/summarize_c3.log 2>&1
fi
( time timeout 1500 python3 bench.py ) > gpurun_out/r06/bench_default_final.log 2>&1
tail -c 600 gpurun_out/r06/bench_default_final.log
