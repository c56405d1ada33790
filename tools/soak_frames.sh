#!/bin/bash
# Soak of the frames/s launcher (frame slots): RUNS runs of FRAMES frames with K frames in flight, no restarts allowed -- every run must
# exit 0 with every frame done.  One line per run and a summary at the end (gpurun_out/soak_<tag>.txt -> profiles/<round>/).
#   bash tools/soak_frames.sh <tag> <runs> <frames> <frames-in-flight> [launcher args...]
set -u
TAG=$1; RUNS=$2; FRAMES=$3; K=$4; shift 4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/soak_$TAG.txt
mkdir -p "$ROOT/gpurun_out"
: > "$OUT"
echo "# python -m vsrd_amd.launcher --gpus 1 --frames $FRAMES --frames-in-flight $K --max-restarts 0 $* : $RUNS runs" | tee -a "$OUT"
fail=0
for run in $(seq 1 "$RUNS"); do
    line=$(cd "$ROOT" && timeout 900 python3 -m vsrd_amd.launcher --gpus 1 --frames "$FRAMES" --frames-in-flight "$K" --max-restarts 0 "$@" 2> /tmp/soak.err | grep '^{' | tail -1)
    code=$?
    if [ -z "$line" ]; then
        fail=$((fail + 1))
        echo "run $run: FAILED (no report line); stderr tail: $(tail -c 600 /tmp/soak.err | tr '\n' ' ')" | tee -a "$OUT"
    else
        python3 - "$run" "$line" <<'PY' | tee -a "$OUT"
import json, sys
d = json.loads(sys.argv[2])
losses = d.get("final_loss_per_frame", {})
print(f"run {sys.argv[1]}: {d['frames']} frames, {d['value']:.3f} frames/s, {d['seconds']:.2f} s, restarts {d['restarts']}, processes {d.get('procs_per_gpu')} x {d.get('frames_in_flight_per_process')} in flight, "
      f"capture s/frame {d['capture_seconds_per_frame']}, "
      f"worst final loss {max(losses.values()) if losses else None:.4f}")
PY
    fi
done
echo "# $RUNS runs, $fail failed" | tee -a "$OUT"
