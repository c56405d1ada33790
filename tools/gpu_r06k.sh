#!/bin/bash
# Round 6 measurement session, second part: the tests that failed, PMC of a batched residual step, regimes, the driver's line.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
( time timeout 900 python -m pytest tests/test_hip_scale.py tests/test_launcher.py -x -q -m gpu -k "config3_full_size_parity or launcher or native or supervisor" 2>&1 | tail -30 ) > gpurun_out/r06/tests_k.log 2>&1
tail -12 gpurun_out/r06/tests_k.log | cut -c1-200
timeout 600 bash tools/pmc_quick.sh native_b16 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" tools/native_mode_bench.py --graph --residual --batch 16 --steps 8 --steps-per-graph 1 > gpurun_out/r06/pmc_native_b16.txt 2>&1
timeout 2400 python3 tools/regimes.py --tag r06 > gpurun_out/r06/regimes.log 2>&1
tail -32 gpurun_out/r06/regimes.log | cut -c1-160
( time timeout 1500 python3 bench.py ) > gpurun_out/r06/bench_default.log 2>&1
tail -c 1500 gpurun_out/r06/bench_default.log
