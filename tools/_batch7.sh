set -u
mkdir -p gpurun_out/r02h
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02h/native_res -o native -- python3 $GRAFT_REPO_ROOT/tools/native_mode_bench.py --graph --residual --steps 200 > $GRAFT_REPO_ROOT/gpurun_out/r02h/native_res.log 2>&1; cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/r02h/native_res.log
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r02h/native_res/native_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} us {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['Percentage']):6.2f}")
PY
rm -f gpurun_out/r02h/native_res/native_kernel_trace.csv
