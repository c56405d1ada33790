set -u
mkdir -p gpurun_out/r02e
python -m pytest tests/test_hip_render.py tests/test_hip_step.py tests/test_hip_dropin.py tests/test_hip_scale.py -m gpu -q -k "residual or graph_mode or unchanged or config3" 2>&1 | tail -25 > gpurun_out/r02e/pytest.log
tail -6 gpurun_out/r02e/pytest.log
python bench.py --residual --views 1 --height 188 --width 704 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02e/trace -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --residual --views 1 --height 188 --width 704 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
head -8 gpurun_out/r02e/trace/bench_kernel_stats.csv | cut -c1-200
python tools/native_mode_bench.py --graph --residual --steps 300 2>&1 | tail -1
