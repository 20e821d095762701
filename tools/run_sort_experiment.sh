cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PTMI_LIBRARY=$R/tools/ab_sort/libptmi.so
for m in 0 1 2 3 4; do
  export PT_EXPERIMENT_SORT=$m
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sortprof$m -o kt -- python3 $R/tools/sort_experiment.py 2>&1 | grep "^sort" || exit 1
  python3 - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/sortprof$m/kt_kernel_stats.csv')))
for r in rows[:6]:
    print('   ', r['Name'][:70], r['Calls'], round(float(r['TotalDurationNs'])/1e6,2), 'ms')
PY
done
