set -u
for c in 1 2 3 4; do
python tools/native_mode_bench.py --graph --residual --steps 200 --concurrent $c 2>&1 | grep "native mode" | cut -c1-200
done
python tools/native_mode_bench.py --graph --steps 400 --concurrent 2 2>&1 | grep "native mode" | cut -c1-200
python tools/native_mode_bench.py --graph --steps 400 --concurrent 4 2>&1 | grep "native mode" | cut -c1-200
