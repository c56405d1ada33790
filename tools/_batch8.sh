for k in 1 2 4 8 16; do echo "slots per item $k"; VSRD_SLOTS_PER_ITEM=$k python tools/native_mode_bench.py --graph --residual --steps 300 2>&1 | tail -1 | cut -c1-110; done
for k in 8 16 32 64; do echo "slots per item $k (C3-shaped)"; VSRD_SLOTS_PER_ITEM=$k python bench.py --residual --views 1 --height 188 --width 704 --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c80-200; done
