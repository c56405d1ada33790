#!/bin/bash
# Time bench.py's headline step with several builds of the library (via gpurun):
#   bash tools/bench_variants.sh <steps> <lib1.so> <lib2.so> ... [-- bench args]
# Every library is a full libvsrd_hip build (VSRD_HIP_LIBRARY selects it); prints one line per library.
set -u
STEPS=$1; shift
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ $# -gt 0 ] && shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/gpurun_out"
for lib in "${LIBS[@]}"; do
    VSRD_HIP_LIBRARY="$ROOT/$lib" python3 "$ROOT/bench.py" --steps "$STEPS" --warmup 3 --no-cpu-baseline --no-extra-regimes "$@" > /tmp/variant.json 2> /tmp/variant.err
    python3 - "$lib" <<'PY' | tee -a "$ROOT/gpurun_out/variants.txt"
import json, sys
try:
    line = [l for l in open("/tmp/variant.json") if l.startswith("{")][-1]
    d = json.loads(line)
    print(f"{sys.argv[1]:48s} {d['ms_per_step']:8.3f} ms/step  {d['value'] / 1e6:8.2f} Mrays/s  loss {d['config'].get('final_loss')}")
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("/tmp/variant.err").read()[-400:])
PY
done
