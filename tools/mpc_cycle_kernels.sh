cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for ee in 0 1; do
python $R/tools/mpc_cycle_profile.py $ee 200
rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/mp$ee -o run -- python $R/tools/mpc_cycle_profile.py $ee 200 > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/mp$ee/run_kernel_stats.csv')))
for r in rows[:14]:
    print(r['Name'][:90].ljust(90), r['Calls'].rjust(6), f"{float(r['AverageNs'])/1e3:8.1f}us", f"{float(r['TotalDurationNs'])/200/1e3:8.1f}us/cycle")
PY
done
