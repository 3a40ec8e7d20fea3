cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 64 128; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/midstats/n$n -- python $R/scripts/dev/mid_run.py $n RBF 60 > /dev/null 2>&1
  f=$(find $R/gpurun_out/midstats/n$n -name "*kernel_stats.csv" | head -1)
  echo "rocprofv3 --kernel-trace --stats, n = $n, evaluated RBF, 60 launches on the 1e6-row grid:"; grep "k_sweep_mid" $f | cut -c1-160
done
