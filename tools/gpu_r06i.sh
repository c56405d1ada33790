cd $GRAFT_REPO_ROOT
( time timeout 3000 python -m pytest tests/test_hip_scale.py -x -q -m gpu -k "config3_full_size_parity" 2>&1 | tail -100 ) > gpurun_out/r06_gputests_5.log 2>&1; tail -8 gpurun_out/r06_gputests_5.log
bash tools/gpu_r06_measure.sh
