set -u
mkdir -p gpurun_out/r02i
for r in 250 1000 4000; do
python tools/native_mode_bench.py --graph --residual --steps 200 --rays $r 2>&1 | grep "native mode" | cut -c1-120
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02i/trace$r -o native -- python3 $GRAFT_REPO_ROOT/tools/native_mode_bench.py --graph --residual --steps 100 --rays $r > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python - $r <<'PY'
import csv, glob, sys
f = glob.glob(f'gpurun_out/r02i/trace{sys.argv[1]}/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:4]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(7), r['AverageNs'].rjust(12), r['Percentage'].rjust(7))
PY
find gpurun_out/r02i/trace$r -name "*.csv" ! -name "*kernel_stats*" -delete
done
