python -m pytest tests -m gpu -x -q 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/st2 -o st -- python $R/bench.py --config 2 --steps 20 --warmup 3 --profile-steps 2 --no-cpu-baseline > $R/gpurun_out/st2.log 2>&1
tail -1 $R/gpurun_out/st2.log | cut -c1-400
python - <<'PY'
import csv,glob,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/st2/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-50s %5s %9.2f"%(r['Name'].replace('(anonymous namespace)::','')[:50], r['Calls'], float(r['AverageNs'])/1e3))
PY
