#!/bin/bash
# rocprofv3 kernel statistics of ONE whole frame of the reference's native mode (3000 steps, hipGraph replay): which kernels the 1.35 s are.
#   bash tools/native_profile.sh [tag] [native_mode_bench args, e.g. --fp32-mlp]   (GPU box, via gpurun; writes gpurun_out/prof_native[_tag]/kernel_stats.txt)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-}; shift || true
OUT=$ROOT/gpurun_out/prof_native${TAG:+_$TAG}
EXTRA="$*"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o native -- python3 "$ROOT/tools/native_mode_bench.py" --graph --whole-frame $EXTRA > "$OUT/stdout.log" 2>&1
cd "$ROOT"
python3 - "$OUT" "$EXTRA" <<'PY'
import csv, glob, sys
out, extra = sys.argv[1], sys.argv[2]
stats = glob.glob(out + '/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(stats[0])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
total = sum(float(r['TotalDurationNs']) for r in rows)
with open(out + '/kernel_stats.txt', 'w') as f:
    f.write('# rocprofv3 --kernel-trace --stats of tools/native_mode_bench.py --graph --whole-frame ' + extra + ' (two frames: one to warm up, one timed)\n')
    f.write(f"{'kernel':84s} {'calls':>8s} {'avg_us':>9s} {'total_ms':>10s} {'pct':>6s}\n")
    for r in rows[:28]:
        f.write(f"{r['Name'].split('(')[0][:84]:84s} {int(r['Calls']):8d} {float(r['AverageNs']) / 1e3:9.2f} {float(r['TotalDurationNs']) / 1e6:10.2f} {100 * float(r['TotalDurationNs']) / total:6.2f}\n")
    f.write('\n# ' + [l for l in open(out + '/stdout.log').read().splitlines() if l.startswith('native mode')][-1] + '\n')
print(open(out + '/kernel_stats.txt').read())
PY
find "$OUT" -type f -size +8M -delete
