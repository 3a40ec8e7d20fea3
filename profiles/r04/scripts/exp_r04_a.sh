#!/bin/bash
# round 4, call A (on the GPU box): where the 4-wave sweep's time goes at config 2, and
# hipEvent vs rocprofv3 at 1x / 4x / 8x the rows.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04a; mkdir -p $OUT; cd $R
python -c "import numpy; print('numpy', numpy.__version__)" > $OUT/env.txt
for m in 0 1 2 4 8 16 32 6 7 15; do
  SAFEOPT_HIP_LIB=scripts/dev/ab/instr.so SGP_ABLATE=$m AB_ONLY=classic AB_TAG="ablate $m" timeout 200 python scripts/dev/ab_sweep.py 2 2>&1 | tail -1
done > $OUT/ablation_cfg2.txt
for mult in 1 4 8; do python scripts/dev/clock_reconcile.py $mult; done > $OUT/reconcile_plain.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for mult in 1 4 8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rp_x$mult -- python $R/scripts/dev/clock_reconcile.py $mult > $OUT/rp_x$mult.log 2>&1
done
cd $R
for mult in 1 4 8; do
  echo "== x$mult"; cat $OUT/rp_x$mult.log | tail -1
  f=$(find $OUT/rp_x$mult -name "*kernel_stats.csv" | head -1); grep -i "k_sweep" $f | head -3
done > $OUT/reconcile_rocprof.txt
cat $OUT/ablation_cfg2.txt $OUT/reconcile_plain.txt $OUT/reconcile_rocprof.txt
