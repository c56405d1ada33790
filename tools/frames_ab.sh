#!/bin/bash
# frames/s A/B inside ONE gpurun call (boxes of the pool differ, and so do two runs on one box): the layouts in turn, the whole turn repeated.
#   bash tools/frames_ab.sh <tag> <frames> <repeats>      -> gpurun_out/frames_ab_<tag>.txt
# layouts: "threads K" = one process, K frames in flight; "procs P x K" = P rank processes on the GPU (--procs-per-gpu P: gloo control plane),
#          K frames in flight each
set -u
TAG=$1; FRAMES=$2; REPEATS=$3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/frames_ab_$TAG.txt
mkdir -p "$ROOT/gpurun_out"; : > "$OUT"
cd "$ROOT"
run() {
    local label=$1; shift
    local line
    line=$(timeout 600 python3 -m vsrd_amd.launcher --frames "$FRAMES" --max-restarts 0 "$@" 2> /tmp/frames_ab.err | grep '^{' | tail -1)
    python3 - "$label" "$line" <<'PY' | tee -a "$OUT"
import json, sys
try:
    d = json.loads(sys.argv[2])
    print(f"{sys.argv[1]:28s} {d['value']:.3f} frames/s  {d['frames']} frames in {d['seconds']:.2f} s  per rank s {[round(x, 2) for x in d.get('per_rank_seconds', [])]}  worst loss {max(d['final_loss_per_frame'].values()):.3f}")
except Exception as error:
    print(f"{sys.argv[1]:28s} FAILED ({error})")
PY
}
for turn in $(seq 1 "$REPEATS"); do
    echo "# turn $turn" | tee -a "$OUT"
    run "threads 1"    --gpus 1 --procs-per-gpu 1 --frames-in-flight 1
    run "threads 3"    --gpus 1 --procs-per-gpu 1 --frames-in-flight 3
    run "threads 5"    --gpus 1 --procs-per-gpu 1 --frames-in-flight 5
    run "procs 2 x 1"  --gpus 1 --procs-per-gpu 2 --frames-in-flight 1
    run "procs 2 x 2"  --gpus 1 --procs-per-gpu 2 --frames-in-flight 2
    run "procs 3 x 1"  --gpus 1 --procs-per-gpu 3 --frames-in-flight 1
done
