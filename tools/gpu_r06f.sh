cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06f
for layout in "--frame-batch 8 --frames 24" "--frame-batch 8 --frames 32" "--frame-batch 16 --frames 32" "--frame-batch 12 --frames 36" "--frame-batch 1 --frames 12"; do
  echo "== $layout"
  timeout 900 python bench.py --native --gpus 1 $layout 2> gpurun_out/r06f/stderr.log | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read())
print({k:l.get(k) for k in ('value','frames','seconds','frame_batch','phase_seconds','slot_setup_seconds')})"
done 2>&1 | tee gpurun_out/r06f/phases.log
