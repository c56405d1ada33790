#!/bin/bash
# Round 6: work items of the MLP adjoint at 32 (default cap) / 48 / 64 slots on config 3 (VSRD_SLOTS_PER_ITEM), two turns.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06q
for turn in 1 2; do
for n in 0 48 64; do
  VSRD_SLOTS_PER_ITEM=$n timeout 600 python3 bench.py --residual --mlp-split-bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-regimes 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  slots per item $n: config 3 split bf16: %.2f ms/step  %.3f Mrays/s  loss %s' % (d['ms_per_step'], d['value']/1e6, d['config']['final_loss']))"
done
done 2>&1 | tee gpurun_out/r06q/timing.log
