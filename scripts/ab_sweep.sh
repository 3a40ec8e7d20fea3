#!/bin/bash
# A/B of the posterior-sweep kernel variants (SGP_SWEEP_VARIANT), interleaved rounds.
# usage: scripts/ab_sweep.sh "0 1 2 3" "2 3" 3
VARIANTS=${1:-"0 1 2 3"}; CONFIGS=${2:-"2 3"}; ROUNDS=${3:-3}
for r in $(seq 1 $ROUNDS); do
  for c in $CONFIGS; do
    for v in $VARIANTS; do
      SGP_SWEEP_VARIANT=$v python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r cfg $c variant $v  sweep_ms %.4f  step_ms %.4f  TF %.2f' % (d['roofline']['kernel_ms_avg'], d['ms_per_step'], d['roofline']['achieved']))"
    done
  done
done
