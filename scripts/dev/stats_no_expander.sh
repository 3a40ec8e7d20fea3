# rocprofv3 --kernel-trace --stats of 20 optimize() calls in the converged config-2-scale state (big passes):
# the per-kernel table -> gpurun_out/kernel_stats_no_expander.csv
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trs
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
ONLY_BIG=1 REPS=20 rocprofv3 --kernel-trace --stats -d /tmp/trs -o trs --output-format csv -- python $GRAFT_REPO_ROOT/scripts/dev/no_expander.py 1000 margin=0.05 ls=0.7 rings=5 dring=0.3 dmid=0.8 dtop=0.4 r0=2.0 dout=1.4 plateau=0.6 > /tmp/trs.log 2>&1
tail -2 /tmp/trs.log | cut -c1-200
F=$(find /tmp/trs -name "*kernel_stats.csv" | head -1)
cp "$F" $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_no_expander.csv
head -30 "$F" | cut -c1-160
