set -u
mkdir -p gpurun_out/r02l
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r02l/pytest.log
tail -3 gpurun_out/r02l/pytest.log
./tools/micro/pk_rate > gpurun_out/r02l/pk_rate.txt 2>&1
python tools/regimes.py --tag r02l > gpurun_out/r02l/regimes.log 2>&1
tail -25 gpurun_out/r02l/regimes.log
bash tools/profile_bench.sh r02 > gpurun_out/r02l/profile_c2.log 2>&1
bash tools/profile_bench.sh r02_c3 --residual --views 1 --height 188 --width 704 > gpurun_out/r02l/profile_c3.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_r02 gpurun_out/r02l/sum_r02
python tools/summarize_profile.py gpurun_out/prof_r02_c3 gpurun_out/r02l/sum_r02_c3
rm -rf gpurun_out/prof_r02 gpurun_out/prof_r02_c3
ls gpurun_out/r02l/sum_r02 gpurun_out/r02l/sum_r02_c3
