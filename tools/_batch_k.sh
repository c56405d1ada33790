set -u
for i in 1 2; do
python tools/native_mode_bench.py --graph --whole-frame 2>&1 | grep "native mode" | cut -c150-330
VSRD_RESIDUAL_WAVE_PER_RAY=1 python tools/native_mode_bench.py --graph --whole-frame 2>&1 | grep "native mode" | cut -c150-330
done
