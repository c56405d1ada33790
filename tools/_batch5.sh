set -u
mkdir -p gpurun_out/r02f
python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/r02f/pytest.log
tail -4 gpurun_out/r02f/pytest.log
python bench.py --residual --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
python tools/native_mode_bench.py --graph --steps 500 2>&1 | tail -1
python tools/native_mode_bench.py --graph --residual --steps 300 --concurrent 2 2>&1 | tail -1
