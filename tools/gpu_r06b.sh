set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06b
timeout 900 python -m pytest tests/test_hip_step.py -x -q -m gpu -k "frame_batch" 2>&1 | tail -25 > gpurun_out/r06b/tests.log
cat gpurun_out/r06b/tests.log
bash tools/native_profile.sh b8 --batch 8 > /dev/null 2>&1
cat gpurun_out/prof_native_b8/kernel_stats.txt
bash tools/native_profile.sh b4 --batch 4 > /dev/null 2>&1
cat gpurun_out/prof_native_b4/kernel_stats.txt
