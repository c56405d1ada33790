set -u
mkdir -p gpurun_out/r02c
python -m pytest tests/test_hip_step.py -m gpu -q 2>&1 | tail -40 > gpurun_out/r02c/pytest_step.log
tail -5 gpurun_out/r02c/pytest_step.log
./tools/micro/mfma_overlap > gpurun_out/r02c/mfma_overlap.log 2>&1
cat gpurun_out/r02c/mfma_overlap.log
python tools/native_mode_bench.py --graph --steps 500 --json > gpurun_out/r02c/native_box_graph.log 2>&1; tail -2 gpurun_out/r02c/native_box_graph.log | head -1
python tools/native_mode_bench.py --graph --residual --steps 300 --json > gpurun_out/r02c/native_res_graph.log 2>&1; tail -2 gpurun_out/r02c/native_res_graph.log | head -1
bash tools/profile_bench.sh r02 > gpurun_out/r02c/profile_c2.log 2>&1
bash tools/profile_bench.sh r02_c3 --residual --views 1 --height 188 --width 704 > gpurun_out/r02c/profile_c3.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_r02 gpurun_out/r02c/sum_r02
python tools/summarize_profile.py gpurun_out/prof_r02_c3 gpurun_out/r02c/sum_r02_c3
rm -rf gpurun_out/prof_r02 gpurun_out/prof_r02_c3
ls gpurun_out/r02c/sum_r02 gpurun_out/r02c/sum_r02_c3
