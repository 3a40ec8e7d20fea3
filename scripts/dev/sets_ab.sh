cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in 0 1; do
  if [ $v = 1 ]; then export SGP_SETS_COPY=1; else unset SGP_SETS_COPY; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-check-chosen --no-shared-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('copy=$v', d['ms_per_step'], d['sets_roofline']['ms'], d.get('bo_iteration',{}).get('optimize_ms'))"
done; done
unset SGP_SETS_COPY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sets_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline --no-check-chosen --no-shared-pass > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/sets_prof -name "*kernel_stats.csv" | head -1); cut -c1-60,200- $f | head -16 ; awk -F'","' '{print $1, $4}' $f | cut -c1-90 | head -16
