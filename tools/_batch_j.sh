set -u
mkdir -p gpurun_out/r02j
python -m pytest tests/test_hip_step.py tests/test_launcher.py -m gpu -q 2>&1 | tail -30 > gpurun_out/r02j/pytest_step.log
grep -n "^E  \|passed\|failed" gpurun_out/r02j/pytest_step.log | cut -c1-250 | head -20
python tools/native_mode_bench.py --graph --residual --steps 300 2>&1 | grep "native mode" | cut -c1-130
python tools/native_mode_bench.py --graph --steps 500 2>&1 | grep "native mode" | cut -c1-130
python tools/native_mode_bench.py --graph --whole-frame 2>&1 | grep "native mode" | cut -c1-330
