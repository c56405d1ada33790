set -u
mkdir -p gpurun_out/r02d
python -m pytest tests/test_hip_render.py tests/test_hip_step.py tests/test_hip_dropin.py -m gpu -q -k "residual or prologue or checkpoint or graph_mode or unchanged" 2>&1 | tail -15 > gpurun_out/r02d/pytest.log
tail -4 gpurun_out/r02d/pytest.log
python tools/phase_timers.py --residual --views 1 --height 188 --width 704 2>&1 | grep -v Warn | tail -11
python bench.py --residual --views 1 --height 188 --width 704 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
