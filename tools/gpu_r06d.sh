cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06d
timeout 900 python -m pytest tests/test_hip_step.py -x -q -m gpu -k "frame_batch" 2>&1 | tail -5
for layout in "--frame-batch 8 --frames 16" "--frame-batch 8 --frames 32" "--frame-batch 4 --frames 16" "--frame-batch 16 --frames 32" "--procs-per-gpu 2 --frame-batch 1 --frames 12"; do
  echo "== $layout"
  timeout 900 python bench.py --native --gpus 1 $layout 2> gpurun_out/r06d/stderr.log | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read())
print({k:l.get(k) for k in ('value','frames','seconds','frame_batch','procs_per_gpu','control_plane','queue','slot_setup_seconds','per_rank_seconds','mean_final_loss','frames_outside_slots')})"
  tail -3 gpurun_out/r06d/stderr.log
done 2>&1 | tee gpurun_out/r06d/launcher.log
