#!/bin/bash
# Round 6, traffic of the residual step (7-float seeds; jets stored only where a tile was evaluated): residual tests, config 3 timing, native batched step, config 3 counters.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06p
timeout 1200 python3 -m pytest tests/test_hip_render.py tests/test_hip_step.py -q -m gpu -x -k "residual or split_bf16 or batch or mlp" 2>&1 | tail -5 | tee gpurun_out/r06p/tests.log
for turn in 1 2; do
  timeout 600 python3 bench.py --residual --mlp-split-bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-regimes 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  config 3 split bf16: %.2f ms/step  %.3f Mrays/s  loss %s' % (d['ms_per_step'], d['value']/1e6, d['config']['final_loss']))"
done 2>&1 | tee gpurun_out/r06p/timing.log
timeout 600 python3 bench.py --residual --steps 3 --warmup 1 --no-cpu-baseline --no-extra-regimes 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  config 3 fp32 mfma: %.2f ms/step  %.3f Mrays/s  loss %s' % (d['ms_per_step'], d['value']/1e6, d['config']['final_loss']))" 2>&1 | tee -a gpurun_out/r06p/timing.log
timeout 600 python3 tools/native_mode_bench.py --graph --residual --batch 16 --steps 200 2>/dev/null | grep "native mode" | cut -c1-200 | tee -a gpurun_out/r06p/timing.log
PROFILE_STEPS=3 PROFILE_WARMUP=1 timeout 1500 bash tools/profile_bench.sh r06p_c3 --residual --mlp-split-bf16 > gpurun_out/r06p/profile.log 2>&1
tail -3 gpurun_out/r06p/profile.log
