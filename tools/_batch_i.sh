set -u
mkdir -p gpurun_out/r02i
python -m pytest tests/test_hip_step.py -m gpu -q 2>&1 | tail -60 > gpurun_out/r02i/pytest_step.log
grep -n "^E  \|passed\|failed" gpurun_out/r02i/pytest_step.log | cut -c1-250 | head -20
