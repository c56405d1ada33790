#!/bin/bash
# Round 6 measurement session (one gpurun call): headline profile + counters, native profiles (one frame / a batch of 16), regimes, the driver's line.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
bash tools/profile_bench.sh r06 > gpurun_out/r06/profile_bench.log 2>&1
python3 tools/summarize_profile.py gpurun_out/prof_r06 gpurun_out/r06/profile_summary > gpurun_out/r06/summarize.log 2>&1
bash tools/native_profile.sh b16 --batch 16 > /dev/null 2>&1
bash tools/native_profile.sh b1 > /dev/null 2>&1
bash tools/pmc_quick.sh native_b16 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" tools/native_mode_bench.py --graph --whole-frame --batch 16 > gpurun_out/r06/pmc_native_b16.txt 2>&1
python3 tools/regimes.py --tag r06 > gpurun_out/r06/regimes.log 2>&1
( time python3 bench.py ) > gpurun_out/r06/bench_default.log 2>&1
tail -c 3000 gpurun_out/r06/bench_default.log
tail -30 gpurun_out/r06/regimes.log
