#!/bin/bash
# One PMC pass over an arbitrary python command (via gpurun): bash tools/pmc_quick.sh <tag> "<counters>" <script.py> [args...]
set -u
TAG=$1; COUNTERS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc $COUNTERS --kernel-trace --output-format csv -d "$OUT" -o run -- python3 "$ROOT/$1" "${@:2}" > "$OUT/stdout.log" 2>&1
tail -n 3 "$OUT/stdout.log" | cut -c1-400
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "vsrd::" in r["Kernel_Name"] or "vsrd_split::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):3d} last={v[-1]:.6g}")
PY
find "$OUT" -type f -size +8M -delete
