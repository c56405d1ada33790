set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06a
timeout 900 python -m pytest tests/test_hip_step.py -x -q -m gpu -k "frame_batch or frame_slot or run_replays or frame_prologue or hypernetwork_kernels" 2>&1 | tail -25 > gpurun_out/r06a/tests.log
cat gpurun_out/r06a/tests.log
for b in 0 1 2 4 8; do
  if [ $b = 0 ]; then timeout 300 python tools/native_mode_bench.py --graph --whole-frame --json 2>&1 | tail -2; else timeout 600 python tools/native_mode_bench.py --graph --whole-frame --batch $b --json 2>&1 | tail -2; fi
done > gpurun_out/r06a/native_batch.log 2>&1
cat gpurun_out/r06a/native_batch.log
