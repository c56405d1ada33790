#!/bin/bash
# Kernel timeline of ONE replayed step of the reference's native mode (1000 rays x 100 samples, residual phase), from a rocprofv3
# kernel trace: start / end / duration / queue of every kernel between two frame_epilogue launches.  GPU box, via gpurun:
#   bash tools/native_timeline.sh [box-only] [extra native_mode_bench.py flags]  > gpurun_out/native_step_timeline.txt
set -u
PHASE=--residual
if [ "${1:-}" = "box-only" ]; then PHASE=; shift; fi
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$(mktemp -d /tmp/native_timeline.XXXXXX)
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o native -- python3 "$ROOT/tools/native_mode_bench.py" --graph $PHASE --steps 40 "$@" > /dev/null 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys
trace = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(rows) if 'frame_epilogue' in r['Kernel_Name']]
a, b = ends[-3], ends[-2]
t0 = int(rows[a]['End_Timestamp'])
print(f"# one replayed step (epilogue to epilogue): {(int(rows[b]['End_Timestamp']) - t0) / 1e3:.1f} us under the profiler")
print("#  start_us    end_us  duration_us  queue  kernel")
for r in rows[a + 1:b + 1]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print(f"{s / 1e3:10.1f} {e / 1e3:9.1f} {(e - s) / 1e3:12.1f}  q{r.get('Queue_Id', '?'):>3s}  {r['Kernel_Name'].split('(')[0][:70]}")
PY
rm -rf "$OUT"
