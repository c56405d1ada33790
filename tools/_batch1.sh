set -u
mkdir -p gpurun_out/r02b
python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r02b/pytest.log
tail -3 gpurun_out/r02b/pytest.log
python tools/phase_timers.py --residual --views 1 --height 188 --width 704 > gpurun_out/r02b/phases_c3.log 2>&1
python tools/phase_timers.py > gpurun_out/r02b/phases_c2.log 2>&1
python tools/phase_timers.py --schedule start > gpurun_out/r02b/phases_c2_start.log 2>&1
python tools/phase_timers.py --schedule end > gpurun_out/r02b/phases_c2_end.log 2>&1
tail -12 gpurun_out/r02b/phases_c3.log gpurun_out/r02b/phases_c2.log
python bench.py --residual --views 1 --height 188 --width 704 --two-launch --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02b/c3_two_launch.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r02b/bench_c2.log 2>&1
python bench.py --residual --views 1 --height 188 --width 704 --steps 5 --warmup 1 > gpurun_out/r02b/bench_c3s.log 2>&1
tail -1 gpurun_out/r02b/bench_c2.log | cut -c1-1500
tail -1 gpurun_out/r02b/bench_c3s.log | cut -c1-1500
