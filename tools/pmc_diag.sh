#!/bin/bash
# Stall diagnosis of the benchmark's step (via gpurun): instruction-cache behaviour, memory latencies by class, branch / fetch counts.
#   bash tools/pmc_diag.sh <tag> [bench args...]
# One rocprofv3 --pmc pass per counter group (the latency metrics accumulate a LEVEL counter and want a pass of their own).
set -u
TAG=${1:-diag}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-extra-regimes $*"
pass() {
    local name=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o run -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$name.log" 2>&1
}
pass icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass ifetch_latency InstrFetchLatency
pass smem_latency SmemLatency
pass vmem_latency VmemLatency
pass lds_latency LdsLatency
pass mem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_ANY
pass dcache SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_TC_STALL SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU
cd "$ROOT"
find "$OUT" -type f -size +8M -delete
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(path)):
        if "vsrd::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[1] + "/summary.txt", "w") as out:
    for k, cs in acc.items():
        print(k, file=out)
        for c, v in sorted(cs.items()):
            print(f"   {c:32s} n={len(v):3d} last={v[-1]:.6g}", file=out)
print(open(sys.argv[1] + "/summary.txt").read())
PY
