set -u
mkdir -p gpurun_out/r02h
python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/r02h/pytest_step.log
tail -12 gpurun_out/r02h/pytest_step.log
python tools/native_mode_bench.py --graph --residual --steps 300 --json > gpurun_out/r02h/native_res_graph.log 2>&1; tail -3 gpurun_out/r02h/native_res_graph.log
python tools/native_mode_bench.py --graph --steps 500 --json > gpurun_out/r02h/native_box_graph.log 2>&1; tail -2 gpurun_out/r02h/native_box_graph.log
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02h/trace -o native -- python3 $GRAFT_REPO_ROOT/tools/native_mode_bench.py --graph --residual --steps 100 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r02h/trace/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:22]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(7), r['AverageNs'].rjust(12), r['Percentage'].rjust(7))
PY
find gpurun_out/r02h/trace -name "*.csv" ! -name "*kernel_stats*" -delete
