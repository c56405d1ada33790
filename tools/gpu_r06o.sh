#!/bin/bash
# the C2 / C5 leads as TIMING BOUNDS (each variant is the shipped kernel with the cost in question removed -- wrong
# results, honest clock): what the lead could buy if its own machinery were free (profiles/r06/variants.txt).
#   freesel    -DVSRD_BOUND_FREE_SELECTORS      per-instance phase as if the seven selectors of a pair cost nothing (selector bits carried from the forward sweep)
#   finer1/2   -DVSRD_BOUND_FINER_CULLING=1|2   every round with >= 5 candidate instances loses 1 | 2 of them (an 8 x 8 mapping's finer culling granularity)
#   cachedmix  -DVSRD_BOUND_CACHED_LABEL_MIX    the reverse sweep's label mix of EVERY round at the price of the cached round's (C5)
#   first: bash tools/build_variant.sh freesel -DVSRD_BOUND_FREE_SELECTORS; ... finer1 -DVSRD_BOUND_FINER_CULLING=1; finer2 -DVSRD_BOUND_FINER_CULLING=2; cachedmix -DVSRD_BOUND_CACHED_LABEL_MIX
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06o
line() { python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  %-10s %-9s %8.3f ms/step  %8.3f Mrays/s' % ('$1', '$2', d['ms_per_step'], d['value']/1e6))"; }
for turn in 1 2; do
  for v in "" _freesel _finer1 _finer2 _cachedmix; do
    VSRD_HIP_LIBRARY=$GRAFT_REPO_ROOT/vsrd_amd/lib/libvsrd_hip$v.so timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-regimes 2>/dev/null | line config2 ${v:-base}
  done
  for v in "" _freesel _finer1 _finer2 _cachedmix; do
    VSRD_HIP_LIBRARY=$GRAFT_REPO_ROOT/vsrd_amd/lib/libvsrd_hip$v.so timeout 300 python3 bench.py --views 17 --height 752 --width 2816 --instances 64 --samples 128 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-regimes 2>/dev/null | line config5 ${v:-base}
  done
done 2>&1 | tee gpurun_out/r06o/timing.log
