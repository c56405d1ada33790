#!/bin/bash
# VERDICT r05 item 3, variants of the split-bf16 products, compiled and timed (profiles/r06_c3_bf16/variants.txt):
#   trunc   -DVSRD_SPLIT_TRUNCATE  both parts of a split truncated to bfloat16 (v_perm_b32 instead of v_cvt_pk_bf16_f32)
#   outer3  -DVSRD_SPLIT_OUTER_3   the weight adjoint's outer products without the lo.lo partial product
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06m
for turn in 1 2; do
for v in "" _trunc _outer3; do
  lib=vsrd_amd/lib/libvsrd_hip$v.so
  echo "== turn $turn ${v:-base}"
  VSRD_HIP_LIBRARY=$GRAFT_REPO_ROOT/$lib timeout 600 python3 bench.py --residual --mlp-split-bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-regimes 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  config 3 split bf16: %.2f ms/step  %.3f Mrays/s  loss %s' % (d['ms_per_step'], d['value']/1e6, d['config']['final_loss']))"
  VSRD_HIP_LIBRARY=$GRAFT_REPO_ROOT/$lib timeout 600 python3 tools/native_mode_bench.py --graph --residual --batch 16 --steps 200 2>/dev/null | grep "native mode" | cut -c1-200
done
done 2>&1 | tee gpurun_out/r06m/timing.log
for v in "" _trunc _outer3; do
  echo "== goldens ${v:-base}"
  VSRD_HIP_LIBRARY=$GRAFT_REPO_ROOT/vsrd_amd/lib/libvsrd_hip$v.so timeout 900 python3 -m pytest tests/test_hip_render.py tests/test_hip_step.py -q -m gpu -k "split_bf16 and (golden or oracle)" 2>&1 | grep "passed\|failed\|vs golden\|labels\|grad " | cut -c1-150 | tail -25
done 2>&1 | tee gpurun_out/r06m/goldens.log
