set -u
mkdir -p gpurun_out/r02g
python -m pytest tests/test_hip_render.py -m gpu -q -k "split_and_single" 2>&1 | tail -4
bash tools/profile_bench.sh r02 > gpurun_out/r02g/profile_c2.log 2>&1
bash tools/profile_bench.sh r02_c3 --residual --views 1 --height 188 --width 704 > gpurun_out/r02g/profile_c3.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_r02 gpurun_out/r02g/sum_r02
python tools/summarize_profile.py gpurun_out/prof_r02_c3 gpurun_out/r02g/sum_r02_c3
rm -rf gpurun_out/prof_r02 gpurun_out/prof_r02_c3
python tools/phase_timers.py > gpurun_out/r02g/phases_c2.log 2>&1
python tools/regimes.py --tag r02g > gpurun_out/r02g/regimes.log 2>&1
tail -25 gpurun_out/r02g/regimes.log
