#!/bin/bash
# Build an experiment variant of the library next to the product build (same ABI; VSRD_HIP_LIBRARY=<path> selects it):
#   bash tools/build_variant.sh <name> [-DMACRO ...]      ->  vsrd_amd/lib/libvsrd_hip_<name>.so
# Same flags as __graft_entry__.HIPCC_FLAGS plus the macros.  The product path never loads these (git-ignored like every .so).
set -eu
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
FLAGS=$(python3 -c "import sys; sys.path.insert(0, '$ROOT'); import __graft_entry__ as g; print(' '.join(g.HIPCC_FLAGS))")
/opt/rocm/bin/hipcc $FLAGS "$@" -o "$ROOT/vsrd_amd/lib/libvsrd_hip_$NAME.so" "$ROOT/vsrd_amd/csrc/api.hip" 2> "/tmp/build_variant_$NAME.log" || { tail -20 "/tmp/build_variant_$NAME.log"; exit 1; }
ls -la "$ROOT/vsrd_amd/lib/libvsrd_hip_$NAME.so"
