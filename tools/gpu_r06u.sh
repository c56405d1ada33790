#!/bin/bash
# Which half of -DVSRD_CULL_INNER breaks config 5's hard bound: the bound test's limit or the exact test's starting minimum?
#   first: bash tools/build_variant.sh innerlimit -DVSRD_CULL_INNER -DVSRD_CULL_INNER_LIMIT_ONLY; bash tools/build_variant.sh innerbest -DVSRD_CULL_INNER -DVSRD_CULL_INNER_BEST_ONLY
#   (the two halves were macros of field.h: cull_round during the experiment; see docs/OPTLOG.md round 6 item 7b -- they are not in the tree any more)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06u
for v in innerlimit innerbest; do
  echo "== $v"
  VSRD_HIP_LIBRARY=$GRAFT_REPO_ROOT/vsrd_amd/lib/libvsrd_hip_$v.so timeout 600 python3 -m pytest tests/test_hip_scale.py -q -m gpu -k "test_full_size_parity_against_the_oracle and config5" 2>&1 | grep -E "^E   +assert|passed|failed|pass 2, determinate" | head -6 | cut -c1-200
  VSRD_HIP_LIBRARY=$GRAFT_REPO_ROOT/vsrd_amd/lib/libvsrd_hip_$v.so timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-regimes 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  config2 %8.3f ms/step  %8.3f Mrays/s' % (d['ms_per_step'], d['value']/1e6))"
done 2>&1 | tee gpurun_out/r06u/log.txt
