cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06c
for spi in 0 8 16 32; do
  for b in 4 8; do
    echo "== VSRD_SLOTS_PER_ITEM=$spi batch $b"
    VSRD_SLOTS_PER_ITEM=$spi timeout 600 python tools/native_mode_bench.py --graph --whole-frame --batch $b 2>&1 | grep "native mode" | cut -c1-420
  done
done > gpurun_out/r06c/slots_per_item.log 2>&1
cat gpurun_out/r06c/slots_per_item.log
