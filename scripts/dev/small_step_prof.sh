cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o ss -- python $GRAFT_REPO_ROOT/scripts/dev/small_step_time.py 2>&1 | tail -3
find /tmp/prof -name "*.csv" | head
python - <<PY
import csv,glob
for f in glob.glob("/tmp/prof/**/*kernel_stats*.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:10]:
        print(r["Name"][:80], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
