#!/bin/bash
# Round 6: -DVSRD_CULL_INNER (the ball every box contains around its centre tightens the exact test's running minimum and the bound test) against the shipped
# library, config 2 / config 5 / two-launch config 2, two turns; then the box-only GPU tests under the variant.
#   first (here, no GPU needed): bash tools/build_variant.sh inner -DVSRD_CULL_INNER
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06t
line() { python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  %-10s %-9s %8.3f ms/step  %8.3f Mrays/s  loss %s' % ('$1', '$2', d['ms_per_step'], d['value']/1e6, d['config']['final_loss']))"; }
for turn in 1 2; do
  for v in "" _inner; do
    VSRD_HIP_LIBRARY=$GRAFT_REPO_ROOT/vsrd_amd/lib/libvsrd_hip$v.so timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-regimes 2>/dev/null | line config2 ${v:-base}
  done
  for v in "" _inner; do
    VSRD_HIP_LIBRARY=$GRAFT_REPO_ROOT/vsrd_amd/lib/libvsrd_hip$v.so timeout 300 python3 bench.py --views 17 --height 752 --width 2816 --instances 64 --samples 128 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-regimes 2>/dev/null | line config5 ${v:-base}
  done
  for v in "" _inner; do
    VSRD_HIP_LIBRARY=$GRAFT_REPO_ROOT/vsrd_amd/lib/libvsrd_hip$v.so timeout 300 python3 bench.py --two-launch --steps 10 --warmup 3 --no-cpu-baseline --no-extra-regimes 2>/dev/null | line two-launch ${v:-base}
  done
done 2>&1 | tee gpurun_out/r06t/timing.log
VSRD_HIP_LIBRARY=$GRAFT_REPO_ROOT/vsrd_amd/lib/libvsrd_hip_inner.so timeout 1500 python3 -m pytest tests/test_hip_render.py tests/test_hip_step.py tests/test_hip_scale.py -q -m gpu -k "not residual and not config3 and not split_bf16 and not mlp" 2>&1 | tail -15 | cut -c1-200 | tee gpurun_out/r06t/tests.log
