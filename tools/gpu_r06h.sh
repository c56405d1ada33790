cd $GRAFT_REPO_ROOT
( time timeout 3000 python -m pytest tests/test_hip_scale.py tests/test_hip_step.py tests/test_launcher.py -x -q -m gpu --deselect "tests/test_hip_scale.py::test_full_size_parity_against_the_oracle" 2>&1 | tail -150 ) > gpurun_out/r06_gputests_4.log 2>&1; tail -8 gpurun_out/r06_gputests_4.log
