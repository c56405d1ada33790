#!/bin/bash
# Profile the benchmark on the GPU box: kernel-trace stats first, then PMC counters in their own passes
# (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).  Usage (via gpurun):
#   bash tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
# the kernel-trace pass runs the driver's own step counts (bench.py --steps 20 --warmup 5 for config 2; PROFILE_STEPS / PROFILE_WARMUP for
# the long regimes), so that the committed average duration is of the run the driver times
ARGS="--steps ${PROFILE_STEPS:-20} --warmup ${PROFILE_WARMUP:-5} --no-cpu-baseline --no-extra-regimes $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python3 "$ROOT/bench.py" $ARGS > "$OUT/trace_stdout.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
  --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o bench -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extra-regimes $* > "$OUT/pmc_sq_stdout.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o bench -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extra-regimes $* > "$OUT/pmc_fetch_stdout.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o bench -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extra-regimes $* > "$OUT/pmc_write_stdout.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES \
  --kernel-trace --output-format csv -d "$OUT/pmc_lds" -o bench -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extra-regimes $* > "$OUT/pmc_lds_stdout.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d "$OUT/pmc_mix" -o bench -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extra-regimes $* > "$OUT/pmc_mix_stdout.log" 2>&1
cd "$ROOT"
for f in "$OUT"/*_stdout.log; do echo "== $f"; tail -n 4 "$f" | cut -c1-600; done
find "$OUT" -type f | head -60
find "$OUT" -type f -size +8M -delete
du -sh "$OUT"
