#!/bin/bash
# last row block with 5..8 rows as two narrow row blocks (SGP_NO_NARROW=2: only the one-block form)
cd "$(dirname "$0")/../.."
OUT=gpurun_out/exp_r02_i
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "predict or small or posterior or full_size" 2>&1 | tail -6 | tee $OUT/pytest.txt
for c in 2 4 3 5; do
  for r in 1 2; do
  echo -n "narrow<=4 only: "; SGP_NO_NARROW=2 timeout 120 python scripts/ablate.py $c 8 2>&1 | tail -1
  echo -n "narrow<=8     : "; timeout 120 python scripts/ablate.py $c 8 2>&1 | tail -1
  done
done | tee $OUT/narrow2.txt
for v in 2 0; do
  echo "SGP_NO_NARROW=$v bench 5:"; SGP_NO_NARROW=$v python bench.py --config 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline'])"
done | tee $OUT/bench5.txt
