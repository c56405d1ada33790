#!/bin/bash
# Build an experiment variant of the library next to the product build (same ABI; VSRD_HIP_LIBRARY=<path> selects it):
#   bash tools/build_variant.sh <name> [-DMACRO ...]      ->  vsrd_amd/lib/libvsrd_hip_<name>.so
# Same translation units and flags as __graft_entry__.build() plus the macros (VSRD_SCHED=default|<strategy> for the scheduler A/B).
# The product path never loads these (git-ignored like every .so).
set -eu
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
python3 -c "import sys, __graft_entry__ as g; g.compile_library('$ROOT/vsrd_amd/lib/libvsrd_hip_$NAME.so', sys.argv[1:])" "$@" 2> "/tmp/build_variant_$NAME.log" || { grep -E "error" -A3 "/tmp/build_variant_$NAME.log" | head -30; exit 1; }
ls -la "$ROOT/vsrd_amd/lib/libvsrd_hip_$NAME.so"
