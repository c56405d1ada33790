cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06e
for layout in "--frame-batch 8 --frames 24" "--frame-batch 1 --adjoint-item-slots 16 --frames 24" "--frame-batch 16 --frames 24" "--frame-batch 1 --frames 24"; do
  echo "== $layout"
  timeout 900 python bench.py --native --gpus 1 $layout 2> gpurun_out/r06e/stderr.log | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read())
print({k:l.get(k) for k in ('value','frames','seconds','frame_batch')})
print(' '.join(f'{k}:{v:.4f}' for k,v in sorted(l['final_loss_per_frame'].items(), key=lambda kv:int(kv[0]))))"
done 2>&1 | tee gpurun_out/r06e/losses.log
