cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06g
( time timeout 3000 python -m pytest tests/test_hip_scale.py -x -q -m gpu -k "config3_full_size_parity" 2>&1 | tail -120 ) > gpurun_out/r06g/parity.log 2>&1
grep -v "^  test_\|^$" gpurun_out/r06g/parity.log | tail -40
grep "determinate\|step's samples\|real" gpurun_out/r06g/parity.log
